/*
 * canvas_hip.h -- C-ABI of the MI355X (gfx950) implementation of Canvas's per-pixel video path.
 *
 * One shared library, libcanvas_hip.so, exports
 *   (1) the symbol set of the reference's src/cprocess for this path, with the same names,
 *       argument meaning and error behaviour, working on HOST frames (each call stages the
 *       frames to HBM, runs the HIP kernels, and copies the result back: drop-in, PCIe-bound);
 *   (2) `cvs_*_dev` twins of the same operations on frames that already live in HBM
 *       (frame->data is a device pointer), taking a stream: zero-copy, what bench.py measures;
 *   (3) the fused colour-matrix + alpha-over chain, single and batched.
 * There is no CPU implementation behind any of these: without a HIP device every pixel entry
 * point fails (empty current_window / non-zero status, message in cvs_last_error()).
 *
 * Struct layouts, field order and names are those of the reference's include/framework.h so
 * that code compiled against it keeps working:
 *   rational, v2i, box2i, v2f, box2f ........ framework.h:46-75
 *   box2i_* / min / max / clamp helpers ..... framework.h:77-149
 *   rgba_f16, rgba_u8, rgba_f32, frames ..... framework.h:155-177
 *   video_frame_source_funcs, video_source .. framework.h:185-194, 210-213
 *   fir_filter .............................. framework.h:618-627
 * GL types are gone: slot 3 of the vtable (get_frame_gl, framework.h:193) keeps its position
 * and becomes the device-frame entry, announced by VIDEO_SOURCE_FLAG_DEVICE in `flags`
 * (framework.h:190 reserves that word).
 */
#ifndef CANVAS_HIP_H
#define CANVAS_HIP_H

#include <stdint.h>
#include <stddef.h>
#include <stdbool.h>

#if defined(__cplusplus)
extern "C" {
#endif

#define CVS_EXPORT __attribute__((visibility("default")))
#define NS_PER_SEC INT64_C(1000000000)

/* ------------------------------------------------------------------ value types */

typedef uint16_t half;                 /* include/half.h:26 */
#define HALF_COUNT 65536

typedef struct { int32_t n; uint32_t d; } rational;
typedef struct { int32_t x, y; } v2i;
typedef struct { v2i min, max; } box2i;        /* inclusive; empty when max < min on either axis */
typedef struct { float x, y; } v2f;
typedef struct { v2f min, max; } box2f;

typedef struct { half r, g, b, a; } rgba_f16;  /* 8 bytes, channel order r,g,b,a */
typedef struct { uint8_t r, g, b, a; } rgba_u8;
typedef struct { float r, g, b, a; } rgba_f32; /* 16 bytes */

/* A frame is a caller-owned buffer covering full_window (rows packed, stride = its width);
 * the callee fills some sub-rectangle and reports it in current_window.  Pixels outside
 * current_window are undefined. */
typedef struct { rgba_f16 *data; box2i full_window; box2i current_window; } rgba_frame_f16;
typedef struct { rgba_f32 *data; box2i full_window; box2i current_window; } rgba_frame_f32;

/* Device-resident frame handed through vtable slot 3. */
enum { CVS_FORMAT_F16 = 1, CVS_FORMAT_F32 = 2 };
typedef struct {
    void *data;            /* device pointer, layout as the host frame of `format` */
    int format;            /* set by the caller: which layout `data` has room for */
    box2i full_window;
    box2i current_window;
    void *stream;          /* hipStream_t the callee must enqueue on (NULL = the library's stream) */
} rgba_frame_dev;

typedef void (*video_get_frame_func)(void *self, int frame_index, rgba_frame_f16 *frame);
typedef void (*video_get_frame_32_func)(void *self, int frame_index, rgba_frame_f32 *frame);
typedef void (*video_get_frame_dev_func)(void *self, int frame_index, rgba_frame_dev *frame);

#define VIDEO_SOURCE_FLAG_DEVICE 0x1   /* slot 3 is a device-frame entry */

typedef struct {
    int flags;
    video_get_frame_func get_frame;
    video_get_frame_32_func get_frame_32;
    video_get_frame_dev_func get_frame_dev;     /* was get_frame_gl */
} video_frame_source_funcs;

typedef struct { void *obj; video_frame_source_funcs *funcs; } video_source;

typedef struct { float *coeff; int width; int center; } fir_filter;

/* ------------------------------------------------------------------ inline helpers (framework.h:55-149,196-208) */

static inline void v2i_add(v2i *r, const v2i *a, const v2i *b) { r->x = a->x + b->x; r->y = a->y + b->y; }
static inline void v2i_subtract(v2i *r, const v2i *a, const v2i *b) { r->x = a->x - b->x; r->y = a->y - b->y; }

#if !defined(__cplusplus)
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline int clamp(int v, int lo, int hi) { return min(max(v, lo), hi); }
#endif
static inline float minf(float a, float b) { return a < b ? a : b; }
static inline float maxf(float a, float b) { return a > b ? a : b; }
static inline float clampf(float v, float lo, float hi) { return minf(maxf(v, lo), hi); }

static inline void box2i_set(box2i *b, int x0, int y0, int x1, int y1) { b->min.x = x0; b->min.y = y0; b->max.x = x1; b->max.y = y1; }
static inline void box2i_set_empty(box2i *b) { box2i_set(b, 0, 0, -1, -1); }
static inline bool box2i_is_empty(const box2i *b) { return b->max.x < b->min.x || b->max.y < b->min.y; }
static inline void box2i_intersect(box2i *r, const box2i *a, const box2i *b) {
    int x0 = a->min.x > b->min.x ? a->min.x : b->min.x, y0 = a->min.y > b->min.y ? a->min.y : b->min.y;
    int x1 = a->max.x < b->max.x ? a->max.x : b->max.x, y1 = a->max.y < b->max.y ? a->max.y : b->max.y;
    box2i_set(r, x0, y0, x1, y1);
}
static inline void box2i_union(box2i *r, const box2i *a, const box2i *b) {
    int x0 = a->min.x < b->min.x ? a->min.x : b->min.x, y0 = a->min.y < b->min.y ? a->min.y : b->min.y;
    int x1 = a->max.x > b->max.x ? a->max.x : b->max.x, y1 = a->max.y > b->max.y ? a->max.y : b->max.y;
    box2i_set(r, x0, y0, x1, y1);
}
/* an inverted axis is turned into the gap between the two spans it came from */
static inline void box2i_normalize(box2i *b) {
    if (b->min.x > b->max.x) { int32_t t = b->min.x - 1; b->min.x = b->max.x + 1; b->max.x = t; }
    if (b->min.y > b->max.y) { int32_t t = b->min.y - 1; b->min.y = b->max.y + 1; b->max.y = t; }
}
static inline void box2i_get_size(const box2i *b, v2i *r) {
    r->x = b->max.x < b->min.x ? 0 : b->max.x - b->min.x + 1;
    r->y = b->max.y < b->min.y ? 0 : b->max.y - b->min.y + 1;
}
static inline rgba_f16 *video_get_pixel_f16(rgba_frame_f16 *f, int x, int y) {
    return &f->data[(ptrdiff_t)(y - f->full_window.min.y) * (f->full_window.max.x - f->full_window.min.x + 1) + x - f->full_window.min.x];
}
static inline rgba_f32 *video_get_pixel_f32(rgba_frame_f32 *f, int x, int y) {
    return &f->data[(ptrdiff_t)(y - f->full_window.min.y) * (f->full_window.max.x - f->full_window.min.x + 1) + x - f->full_window.min.x];
}

/* ------------------------------------------------------------------ (1) reference symbol set, HOST frames
 * Each entry names the reference definition it replaces. */

/* src/cprocess/half.c:87-105 -- function-pointer globals, valid after init_half() */
CVS_EXPORT void init_half(void);
CVS_EXPORT extern void (*half_convert_from_float)(half *out, const float *in, int count);       /* truncating, half.c:47-51 */
CVS_EXPORT extern void (*half_convert_to_float)(float *out, const half *in, int count);         /* exact, half.c:31-37 */
CVS_EXPORT extern void (*half_convert_from_float_fast)(half *out, const float *in, int count);  /* half.c:53-59 */
CVS_EXPORT extern void (*half_convert_to_float_fast)(float *out, const half *in, int count);    /* half.c:39-45 */
CVS_EXPORT extern void (*half_lookup)(const half *table, half *out, const half *in, int count); /* half.c:82-85 */

static inline void rgba_f32_to_f16(rgba_f16 *out, const rgba_f32 *in, int count) { half_convert_from_float(&out->r, &in->r, count * 4); }
static inline void rgba_f16_to_f32(rgba_f32 *out, const rgba_f16 *in, int count) { half_convert_to_float(&out->r, &in->r, count * 4); }

/* src/cprocess/main.c:23-31 */
CVS_EXPORT int64_t get_frame_time(const rational *frame_rate, int frame);
CVS_EXPORT int get_time_frame(const rational *frame_rate, int64_t time);
/* src/cprocess/clock.c:28-52 */
CVS_EXPORT int64_t gettime(void);

/* src/cprocess/main.c:33-76,105-144 -- vtable dispatch + format conversion; NULL source => empty window */
CVS_EXPORT void video_get_frame_f16(video_source *source, int frame_index, rgba_frame_f16 *frame);
CVS_EXPORT void video_get_frame_f32(video_source *source, int frame_index, rgba_frame_f32 *frame);
/* src/cprocess/main.c:78-103,146-172 -- the forced pull through vtable slot 3 that force_gl=True asks for
 * (src/process/RgbaFrameF16.c:247-249): here slot 3 is the device slot; a source without one is pulled the
 * ordinary way instead of yielding an empty window */
CVS_EXPORT void video_get_frame_f16_gl(video_source *source, int frame_index, rgba_frame_f16 *frame);
CVS_EXPORT void video_get_frame_f32_gl(video_source *source, int frame_index, rgba_frame_f32 *frame);
/* device twin: fills a device frame through slot 3 when the source has one, else pulls a host
 * frame and uploads it */
CVS_EXPORT void video_get_frame_dev(video_source *source, int frame_index, rgba_frame_dev *frame);

/* src/cprocess/video_mix.c:27-44, 73-105, 107-235, 46-71, 237-370 */
CVS_EXPORT void video_copy_frame_f16(rgba_frame_f16 *out, rgba_frame_f16 *in);
CVS_EXPORT void video_copy_frame_alpha_f32(rgba_frame_f32 *out, rgba_frame_f32 *in, float alpha);
/* include/framework.h:236 declares video_attenuate_f32 and no reference source defines it; here it is the in-place
 * case of video_copy_frame_alpha_f32 (video_mix.c:73-105 with out == in): alpha 1 leaves the frame alone, alpha 0
 * empties its window, anything between multiplies the alpha channel over the current window. */
CVS_EXPORT void video_attenuate_f32(rgba_frame_f32 *frame, float alpha);
CVS_EXPORT void video_mix_cross_f32(rgba_frame_f32 *out, rgba_frame_f32 *a, rgba_frame_f32 *b, float mix_b);
CVS_EXPORT void video_mix_cross_f32_pull(rgba_frame_f32 *out, video_source *a, int frame_a, video_source *b, int frame_b, float mix_b);
CVS_EXPORT void video_mix_over_f32(rgba_frame_f32 *out, rgba_frame_f32 *b, float mix_b);

/* src/cprocess/video_scale.c:231-286, 288-319 */
CVS_EXPORT void video_scale_bilinear_f32(rgba_frame_f32 *target, v2f target_point, rgba_frame_f32 *source, v2f source_point, v2f factors);
CVS_EXPORT void video_scale_bilinear_f32_pull(rgba_frame_f32 *target, v2f target_point, video_source *source, int frame,
                                              box2i *source_rect, v2f source_point, v2f factors);

/* src/cprocess/gammatab.c:83-110, 128-159, 171-198, 223-250, 13-38 */
CVS_EXPORT void video_transfer_rec709_to_linear_scene(half *out, const half *in, size_t count);
CVS_EXPORT void video_transfer_rec709_to_linear_display(half *out, const half *in, size_t count);
CVS_EXPORT void video_transfer_linear_to_rec709(half *out, const half *in, size_t count);
CVS_EXPORT void video_transfer_linear_to_sRGB(half *out, const half *in, size_t count);
CVS_EXPORT const uint8_t *video_get_gamma45_ramp(void);

/* src/cprocess/color.c:104-137, 140-165 (exported there, missing from framework.h) */
CVS_EXPORT void video_color_rgb_to_xyz_sdtv(rgba_frame_f16 *frame);
CVS_EXPORT void video_color_xyz_to_srgb(rgba_frame_f16 *frame);

/* src/cprocess/filter.c:24-76, 78-148, 150-153 */
CVS_EXPORT void filter_createTriangle(float sub, float offset, fir_filter *filter);
CVS_EXPORT void filter_createLanczos(float sub, int kernel_size, float offset, fir_filter *filter);
CVS_EXPORT void filter_free(fir_filter *filter);

/* src/cprocess/video_filter.c:27-39 + src/cprocess/gl.c:584 -- the reference only has this as a
 * GLSL program (video_filter_gain_offset_gl); same formula and window rule, f16 frames */
CVS_EXPORT void video_filter_gain_offset_f16(rgba_frame_f16 *out, rgba_frame_f16 *in, float gain, float offset);

/* src/process/SolidColorVideoSource.c:52-101 (the fill loops; parameter fetch stays with the caller) */
CVS_EXPORT void video_fill_solid_f16(rgba_frame_f16 *frame, const box2i *window, const rgba_f32 *color);
CVS_EXPORT void video_fill_solid_f32(rgba_frame_f32 *frame, const box2i *window, const rgba_f32 *color);

/* src/cprocess/workspace.c (video half): item bookkeeping :107-492, stack :494-550, source :604-613 */
typedef struct workspace_t_tag workspace_t;
typedef struct workspace_item_t_tag workspace_item_t;
CVS_EXPORT workspace_t *workspace_create(void);
CVS_EXPORT int workspace_get_length(workspace_t *workspace);
CVS_EXPORT workspace_item_t *workspace_add_item(workspace_t *self, void *source, int64_t x, int64_t width, int64_t offset, int64_t z, void *tag);
CVS_EXPORT workspace_item_t *workspace_get_item(workspace_t *self, int index);
CVS_EXPORT void workspace_remove_item(workspace_item_t *item);
CVS_EXPORT void workspace_as_video_source(workspace_t *workspace, video_source *source);
CVS_EXPORT void workspace_free(workspace_t *workspace);
CVS_EXPORT void workspace_get_item_pos(workspace_item_t *item, int64_t *x, int64_t *width, int64_t *z);
CVS_EXPORT int64_t workspace_get_item_offset(workspace_item_t *item);
CVS_EXPORT void workspace_set_item_offset(workspace_item_t *item, int64_t offset);
CVS_EXPORT void *workspace_get_item_source(workspace_item_t *item);
CVS_EXPORT void workspace_set_item_source(workspace_item_t *item, void *source);
CVS_EXPORT void *workspace_get_item_tag(workspace_item_t *item);
CVS_EXPORT void workspace_set_item_tag(workspace_item_t *item, void *tag);
CVS_EXPORT void workspace_update_item(workspace_item_t *item, int64_t *x, int64_t *width, int64_t *z, int64_t *offset, void **source, void **tag);

/* ------------------------------------------------------------------ (2) device runtime + device-frame twins */

typedef void *cvs_stream_t;            /* hipStream_t */

enum { CVS_LUT_NONE = -1, CVS_LUT_REC709_TO_LINEAR_SCENE = 0, CVS_LUT_REC709_TO_LINEAR_DISPLAY = 1,
       CVS_LUT_LINEAR_TO_REC709 = 2, CVS_LUT_LINEAR_TO_SRGB = 3, CVS_LUT_COUNT = 4 };

/* Arithmetic flavour.  The reference has two builds (SConstruct:46-48,75-83): gcc -std=c99, which rounds every multiply and
 * every add on its own, and clang -- preferred when it is installed -- which contracts a * b + c inside one expression into
 * a fused multiply-add: the matrix rows (src/cprocess/color.c:34-42), the blend numerators (src/cprocess/video_mix.c:193-205,
 * 323-337), every FIR accumulation t += s * coeff (src/cprocess/video_scale.c:82-85), the scaler's line centres (:65) and
 * intermediate window (:257-277), two of the transfer functions (src/cprocess/gammatab.c:58-66,201-211).
 *   CVS_ARITH_SEPARATE   (default)  bit-equal to the gcc build;
 *   CVS_ARITH_CONTRACTED            bit-equal to the clang build (half the FIR instructions: the faster of the two).
 * Process-wide; every entry point reads it once on entry, so change it between calls, not while other threads are inside the
 * library.  Tables that depend on it are cached per flavour.  The environment variable CVS_ARITHMETIC=contracted selects the
 * second flavour at start-up.  cvs_set_arithmetic returns the previous mode, or -1 for an unknown one (nothing changes). */
enum { CVS_ARITH_SEPARATE = 0, CVS_ARITH_CONTRACTED = 1 };
CVS_EXPORT int cvs_set_arithmetic(int mode);
CVS_EXPORT int cvs_get_arithmetic(void);

CVS_EXPORT int cvs_init(int device);                   /* 0 on success; idempotent per device */
CVS_EXPORT int cvs_device_count(void);
CVS_EXPORT int cvs_current_device(void);               /* HIP device of the calling thread's context, -1 before any */

/* Device contexts: several GPUs in one process.  A context is one HIP device plus everything the library keeps on it (scratch
 * pool, cached transfer / tap / byte tables, one stream per calling thread).  cvs_init(device) opens -- or finds -- a context
 * on `device` and makes it the DEFAULT: what every thread runs in that never chose another (and all a single-GPU host ever
 * needs).  cvs_context_open(device) opens a further one, on another GPU of the node or on the same one again (two contexts on
 * one device share nothing but the device); its id (0 .. 63) or -1.  cvs_set_context(ctx) binds the calling thread (-1: back to
 * the default) and returns what it was bound to (-2 for an unknown id, nothing changes).  Every entry point runs in the calling
 * thread's context; frames, streams and events belong to the context (device) they were created in.
 * Frames are independent random-access units (docs/sphinx/framework.rst:16-18): a host shards by giving frame g to context
 * cvs_frame_owner(g, n) -- one thread per context, as the reference's pull queue gives frames to its pool threads
 * (src/process/VideoPullQueue.c:99-113); the graph's parameters are host values and every context builds the device tables it
 * needs on first use.  Nothing is exchanged between devices. */
CVS_EXPORT int cvs_context_open(int device);
CVS_EXPORT int cvs_context_count(void);
CVS_EXPORT int cvs_context_device(int ctx);
CVS_EXPORT int cvs_set_context(int ctx);
CVS_EXPORT int cvs_current_context(void);
CVS_EXPORT int cvs_frame_owner(int64_t frame_index, int nowners);        /* frame_index mod nowners (non-negative); -1 for nowners <= 0 */
CVS_EXPORT const char *cvs_last_error(void);
CVS_EXPORT void cvs_clear_last_error(void);                                           /* forget the calling thread's message */
/* Diagnostics: every failure message (the text of cvs_last_error) is also handed to the installed handler, or written to
 * stderr when there is none.  The handler may be called from any thread that calls into the library.  The reference
 * logs through g_log in per-file domains (e.g. src/cprocess/video_reconstruct.c:23-24); here the domain is always
 * "fluggo.media.cprocess". */
enum { CVS_LOG_ERROR = 0, CVS_LOG_WARNING = 1, CVS_LOG_INFO = 2 };
typedef void (*cvs_log_func)(const char *domain, int level, const char *message, void *user_data);
CVS_EXPORT void cvs_set_log_handler(cvs_log_func handler, void *user_data);           /* thread-local, "" when none */
CVS_EXPORT const char *cvs_device_name(void);
CVS_EXPORT int cvs_compute_units(void);

CVS_EXPORT void *cvs_malloc(size_t bytes);
CVS_EXPORT void cvs_free(void *dev);
/* recycled scratch for per-frame intermediates (no device sync on free, unlike hipFree); a block
 * handed to a different stream than it was last used on waits for that stream first */
/* HIP graphs: record a sequence of device-frame calls on a stream once, replay it with one submission (worth it when
 * the sequence is launch-bound: many short kernels on small frames).  Rules: capturing thread = calling thread; only
 * `cvs_*_dev` entry points on the capturing stream in between; the same sequence must have run once before (tables,
 * pool blocks); frame pointers and parameters are baked in, frame CONTENTS are read at replay time; scratch blocks the
 * sequence took from the pool stay with the graph until cvs_graph_destroy. */
typedef void *cvs_graph_t;
CVS_EXPORT int cvs_graph_begin(cvs_stream_t stream);
CVS_EXPORT cvs_graph_t cvs_graph_end(cvs_stream_t stream);          /* NULL on failure */
CVS_EXPORT int cvs_graph_launch(cvs_graph_t graph, cvs_stream_t stream);
CVS_EXPORT void cvs_graph_destroy(cvs_graph_t graph);
CVS_EXPORT int cvs_mem_info(size_t *free_bytes, size_t *total_bytes);     /* HBM free / total on the bound device */
CVS_EXPORT void *cvs_pool_malloc(size_t bytes, cvs_stream_t s);
CVS_EXPORT void cvs_pool_free(void *dev, cvs_stream_t s);
CVS_EXPORT void cvs_pool_trim(void);
CVS_EXPORT int cvs_memcpy_h2d(void *dev, const void *host, size_t bytes, cvs_stream_t s);
CVS_EXPORT int cvs_memcpy_d2h(void *host, const void *dev, size_t bytes, cvs_stream_t s);
CVS_EXPORT int cvs_memcpy_d2d(void *dst, const void *src, size_t bytes, cvs_stream_t s);
CVS_EXPORT int cvs_memset(void *dev, int value, size_t bytes, cvs_stream_t s);
CVS_EXPORT cvs_stream_t cvs_stream_create(void);
CVS_EXPORT void cvs_stream_destroy(cvs_stream_t s);
CVS_EXPORT int cvs_stream_sync(cvs_stream_t s);
/* HIP events on a stream, for callers that time kernels without linking HIP themselves */
typedef void *cvs_event_t;
CVS_EXPORT cvs_event_t cvs_event_create(void);
CVS_EXPORT void cvs_event_destroy(cvs_event_t e);
CVS_EXPORT int cvs_event_record(cvs_event_t e, cvs_stream_t s);
CVS_EXPORT int cvs_event_sync(cvs_event_t e);
CVS_EXPORT float cvs_event_elapsed_ms(cvs_event_t start, cvs_event_t stop);   /* < 0 on error */

/* the four transfer tables, resident in HBM after first use (128 KiB each); host copy for callers
 * that want to ship them elsewhere (multi-GPU parameter broadcast) */
CVS_EXPORT const half *cvs_lut_device(int which);
CVS_EXPORT const half *cvs_lut_host(int which);
/* install a table built elsewhere (rank 0 broadcasts, the others install); 65536 entries */
CVS_EXPORT int cvs_lut_install(int which, const half *table_host);

/* flat arrays in HBM */
CVS_EXPORT int cvs_half_to_float_dev(float *out, const half *in, size_t count, cvs_stream_t s);
CVS_EXPORT int cvs_float_to_half_dev(half *out, const float *in, size_t count, cvs_stream_t s);
CVS_EXPORT int cvs_half_to_float_fast_dev(float *out, const half *in, size_t count, cvs_stream_t s);
CVS_EXPORT int cvs_float_to_half_fast_dev(half *out, const float *in, size_t count, cvs_stream_t s);
CVS_EXPORT int cvs_half_lookup_dev(const half *table_dev, half *out, const half *in, size_t count, cvs_stream_t s);

/* frames in HBM (frame->data is a device pointer); window logic runs on the host exactly as in
 * the reference and the structs are updated before the call returns; pixels are enqueued on `s` */
CVS_EXPORT int cvs_frame_f16_to_f32_dev(rgba_frame_f32 *out, const rgba_frame_f16 *in, cvs_stream_t s);   /* main.c:115-139 */
CVS_EXPORT int cvs_frame_f32_to_f16_dev(rgba_frame_f16 *out, const rgba_frame_f32 *in, cvs_stream_t s);   /* main.c:43-71 */
CVS_EXPORT int cvs_copy_frame_f16_dev(rgba_frame_f16 *out, const rgba_frame_f16 *in, cvs_stream_t s);
CVS_EXPORT int cvs_copy_frame_alpha_f32_dev(rgba_frame_f32 *out, const rgba_frame_f32 *in, float alpha, cvs_stream_t s);
CVS_EXPORT int cvs_mix_cross_f32_dev(rgba_frame_f32 *out, const rgba_frame_f32 *a, const rgba_frame_f32 *b, float mix_b, cvs_stream_t s);
CVS_EXPORT int cvs_mix_over_f32_dev(rgba_frame_f32 *out, const rgba_frame_f32 *b, float mix_b, cvs_stream_t s);
/* colour matrix with the color.c structure; m is column-major as color.c passes it:
 * m[0..2] = coefficients multiplying r (a.x a.y a.z), m[3..5] those of g, m[6..8] those of b */
CVS_EXPORT int cvs_color_matrix_f16_dev(rgba_frame_f16 *frame, const float m[9], int pre_lut, int post_lut, cvs_stream_t s);
/* the same filter, out of place: out.current = out.full ∩ in.current, one pass */
CVS_EXPORT int cvs_color_matrix_f16_to_dev(rgba_frame_f16 *out, const rgba_frame_f16 *in, const float m[9], int pre_lut, int post_lut, cvs_stream_t s);
CVS_EXPORT int cvs_gain_offset_f16_dev(rgba_frame_f16 *out, const rgba_frame_f16 *in, float gain, float offset, cvs_stream_t s);
CVS_EXPORT int cvs_fill_solid_f16_dev(rgba_frame_f16 *frame, const box2i *window, const rgba_f32 *color, cvs_stream_t s);
CVS_EXPORT int cvs_fill_solid_f32_dev(rgba_frame_f32 *frame, const box2i *window, const rgba_f32 *color, cvs_stream_t s);
/* the f16 twin: widen on load, truncate on store (the f16 pull of a scaler node over a half-native source) */
CVS_EXPORT int cvs_scale_bilinear_f16_dev(rgba_frame_f16 *target, v2f tp, const rgba_frame_f16 *source, v2f sp, v2f fac, cvs_stream_t stream);
/* The two device entries of the scaler for `count` INDEPENDENT frames (a pull queue's frames in flight): frames of one geometry
 * whose buffers do not overlap go up to eight at a time into one launch where the tile kernel takes the call (vertical pass
 * first -- video_scale.c:252 -- and short tap lists: enlarging; 1080p -> 4K: 0.0173 -> see DESIGN.md 4.4 per frame); every
 * other combination is carried out frame by frame, exactly as `count` single calls.  Results are those of the single calls,
 * bit for bit; a NULL entry or a source window outside its buffer refuses the whole call before any launch. */
CVS_EXPORT int cvs_scale_bilinear_f32_batch_dev(rgba_frame_f32 *const *targets, v2f target_point, const rgba_frame_f32 *const *sources, v2f source_point,
                                                v2f factors, int count, cvs_stream_t stream);
CVS_EXPORT int cvs_scale_bilinear_f16_batch_dev(rgba_frame_f16 *const *targets, v2f target_point, const rgba_frame_f16 *const *sources, v2f source_point,
                                                v2f factors, int count, cvs_stream_t stream);
CVS_EXPORT int cvs_scale_bilinear_f32_dev(rgba_frame_f32 *target, v2f target_point, const rgba_frame_f32 *source, v2f source_point, v2f factors, cvs_stream_t s);
/* separable FIR blur at factor 1 (absent from the reference; defined in DESIGN.md) and the Lanczos
 * gather resampler built on filter_createLanczos */
CVS_EXPORT int cvs_fir_blur_f32_dev(rgba_frame_f32 *target, const rgba_frame_f32 *source, const float *taps_host, int ntaps, cvs_stream_t s);
/* DV 4:1:1 edge: planar 8-bit Y'CbCr (Y 720x480, Cb and Cr 180x480) <-> half RGBA on the fixed NTSC DV raster
 * whose first line sits at y = -1.  coded_image: include/framework.h:468-476; alloc: video_subsample.c:23-68. */
#define CODED_IMAGE_MAX_PLANES 4
typedef struct {
    void *data[CODED_IMAGE_MAX_PLANES];
    int stride[CODED_IMAGE_MAX_PLANES];
    int line_count[CODED_IMAGE_MAX_PLANES];
    void (*free_func)(void *image);          /* GFreeFunc */
} coded_image;
CVS_EXPORT coded_image *coded_image_alloc(const int *strides, const int *line_counts, int count);
CVS_EXPORT coded_image *coded_image_alloc0(const int *strides, const int *line_counts, int count);
CVS_EXPORT void video_reconstruct_dv(rgba_frame_f16 *frame, coded_image *planar);       /* video_reconstruct.c:50-137 */
CVS_EXPORT coded_image *video_subsample_dv(rgba_frame_f16 *frame);                      /* video_subsample.c:99-187; encodes the frame's rows in place, as there */
/* device frame + device planes (planar->data[] are device pointers) */
CVS_EXPORT int cvs_reconstruct_dv_dev(rgba_frame_f16 *frame, const coded_image *planar, cvs_stream_t stream);
CVS_EXPORT int cvs_subsample_dv_dev(coded_image *planar, rgba_frame_f16 *frame, int encode_input_in_place, cvs_stream_t stream);

/* 2:3 pulldown removal (src/process/Pulldown23RemovalFilter.c:43-107; the reference keeps this inside the Python
 * node, the arithmetic and the field weave are entry points here so that the node is a thin caller).
 * cvs_pulldown23_frames: output frame -> source frame(s) for cadence phase `offset` (:51-71); returns 1 when the
 *   frame is woven from two (odd rows from *first, even rows from *second), else 0 and *first alone.
 * cvs_weave_fields_f16_dev: :88-104 on device frames -- `other` must be allocated for exactly frame->current_window,
 *   as the reference allocates it; the even rows (first even y >= min.y) of the frame are replaced.  The reference
 *   addresses the second buffer from x = 0 instead of min.x (:101); inside the allocation that shift is reproduced,
 *   outside it (and outside other->current_window, uninitialised in the reference) the pixel is zero. */
CVS_EXPORT int cvs_pulldown23_frames(int offset, int frame_index, int *first, int *second);
CVS_EXPORT int cvs_weave_fields_f16_dev(rgba_frame_f16 *frame, const rgba_frame_f16 *other, cvs_stream_t stream);

/* Display / export edge: the current window of an f16 frame as 4 bytes per pixel, packed row by row.
 * pre_lut: a transfer table applied to all four halfs first (CVS_LUT_NONE for none); then a half->u8 ramp.
 *
 * cvs_frame_to_bytes_dev / video_frame_to_bytes use the gamma-0.45 ramp of video_get_gamma45_ramp():
 *   CVS_DISPLAY_RGBA8: bytes r,g,b,a -- with CVS_LUT_NONE the exporter's conversion (src/libav/writeVideo.c:328-340,
 *     whose own ramp at :108-115 holds the same bytes).
 *   CVS_DISPLAY_ARGB32_PREMUL: a<<24 | (r*a>>8)<<16 | (g*a>>8)<<8 | (b*a>>8), what RgbaFrameF16.to_argb32_bytes
 *     returns (src/process/RgbaFrameF16.c:114-149).
 * cvs_frame_to_rgba8_intent_dev / video_frame_to_rgba8_intent use the software widget's ramp,
 *   lrint(clamp(x^rendering_intent * 255, 0, 255)) (src/cprocess/widget_gl.c:955-968): with CVS_LUT_LINEAR_TO_SRGB
 *   and the widget's default intent 1.25 (:428-429) they are its frame conversion (:291-307), bytes r,g,b,a. */
enum { CVS_DISPLAY_RGBA8 = 0, CVS_DISPLAY_ARGB32_PREMUL = 1 };
CVS_EXPORT int cvs_frame_to_bytes_dev(void *dst_dev, const rgba_frame_f16 *frame, int pre_lut, int mode, cvs_stream_t stream);
CVS_EXPORT int video_frame_to_bytes(void *dst_host, const rgba_frame_f16 *host_frame, int pre_lut, int mode);
CVS_EXPORT int cvs_frame_to_rgba8_intent_dev(void *dst_dev, const rgba_frame_f16 *frame, int pre_lut, float rendering_intent, cvs_stream_t stream);
CVS_EXPORT int video_frame_to_rgba8_intent(void *dst_host, const rgba_frame_f16 *host_frame, int pre_lut, float rendering_intent);

/* blur node between two f16 frames (widen on load, f32 passes, truncate on store), one launch */
CVS_EXPORT int cvs_fir_blur_f16_dev(rgba_frame_f16 *target, const rgba_frame_f16 *source, const float *taps, int ntaps, cvs_stream_t stream);
/* f16 pull of a workspace whose lowest item is a blur node on `source` and whose higher items are `overlays`
 * (bottom first); the blur result stays f32 until the final truncation, as workspace.c:530-544 would have it */
CVS_EXPORT int cvs_blur_over_f16_dev(rgba_frame_f16 *out, const rgba_frame_f16 *source, const float *taps, int ntaps,
                                     const rgba_frame_f16 *const *overlays, int noverlays, cvs_stream_t stream);
CVS_EXPORT int cvs_resample_lanczos_f32_dev(rgba_frame_f32 *target, const rgba_frame_f32 *source, float factor_x, float factor_y, int kernel_size, cvs_stream_t s);
/* the same between f16 frames (widen on load, f32 passes, truncate on store): an f16 pull of the node over a half-native source */
CVS_EXPORT int cvs_resample_lanczos_f16_dev(rgba_frame_f16 *target, const rgba_frame_f16 *source, float factor_x, float factor_y, int kernel_size, cvs_stream_t s);
/* BASELINE config 3 on f16 frames: widen -> blur -> Lanczos resample -> truncate, f32 in between, two launches */
CVS_EXPORT int cvs_blur_lanczos_f16_dev(rgba_frame_f16 *target, const rgba_frame_f16 *source, const float *taps_host, int ntaps,
                                        float factor_x, float factor_y, int kernel_size, cvs_stream_t s);
/* The two entries above for `count` INDEPENDENT frames (a pull queue's frames in flight, a render farm's batch): frames
 * of one geometry that do not feed each other go up to eight at a time into one launch, whose row segments are sized for
 * the whole batch and re-filter fewer halo rows per frame (config 3: 0.084 -> 0.067 ms per 4K frame); every other
 * combination is carried out frame by frame, exactly as `count` single calls.  Results are those of the single calls, bit
 * for bit.  overlays: count x noverlays pointers, frame-major. */
CVS_EXPORT int cvs_blur_over_f16_batch_dev(rgba_frame_f16 *const *outs, const rgba_frame_f16 *const *sources, const float *taps, int ntaps,
                                           const rgba_frame_f16 *const *overlays, int noverlays, int count, cvs_stream_t stream);
CVS_EXPORT int cvs_blur_lanczos_f16_batch_dev(rgba_frame_f16 *const *targets, const rgba_frame_f16 *const *sources, int count,
                                              const float *taps_host, int ntaps, float fx, float fy, int ksize, cvs_stream_t stream);

/* The separable FIR entry points choose among four kernels: the register-window kernel (one tap list for
 * every line), and for per-line tables the gather per target line, tiles in LDS, and the sweep with a lane per pixel.
 * All compute the same sums in the same order -- results are bit-equal, which the parity tests show by pinning each kernel
 * in turn through this call.  It changes speed only, never pixels; process-wide; 0 restores the automatic choice. */
enum { CVS_FIR_PATH_AUTO = 0, CVS_FIR_PATH_PASSES = 1 /* no launch that does both passes: the reference's two passes through an f32 frame */, CVS_FIR_PATH_TILED = 2, CVS_FIR_PATH_TABLES = 4 /* skip the register-window kernel */,
       CVS_FIR_PATH_HV = 16 /* per-line gather, horizontal pass first (the automatic first choice) */,
       CVS_FIR_PATH_ONE_COLUMN = 32 /* the register-window kernels (blur, blur + halving) with one column per lane, never two */,
       CVS_FIR_PATH_TWO_COLUMNS = 64 /* ... with two columns per lane wherever that form takes the launch, narrow frames included */,
       CVS_FIR_PATH_STRIPS = 128 /* the vertical-first triangle scaler on k_fir_vh's strips even where its tile form (k_fir_tile_vh) would take the call */,
       CVS_FIR_PATH_TILES = 256 /* ... on the tile form wherever it takes the call (left alone, the library uses it where it measured faster: floats, and halfs up to about a 4K target) */ };
CVS_EXPORT void cvs_fir_path_override(int mode);
/* Which kernel the calling thread's last FIR launch (scaler, blur, Lanczos resample, blur + resample) went to -- what a
 * test pinned to one kernel asserts, and what tells a silent fallback from the intended kernel.  A fused kernel that was
 * chosen and then failed to launch is reported through the log handler before the next one is tried. */
enum { CVS_FIR_KERNEL_NONE = 0,
       CVS_FIR_KERNEL_WINDOW = 1,      /* k_blur: one tap list for every line, vertical window in registers */
       CVS_FIR_KERNEL_HALVE = 2,       /* k_blur_halve: blur + Lanczos halving in one sweep */
       /* 3: the channel-pair sweep of rounds 2-3 (k_fir_lanes), retired in round 4 in favour of k_fir_hv */
       CVS_FIR_KERNEL_VH = 4,          /* k_fir_vh: the triangle scaler with the vertical pass first */
       CVS_FIR_KERNEL_TILED = 5,       /* k_fir2d: LDS tiles */
       CVS_FIR_KERNEL_RETIRED_6 = 6,   /* (k_fir_stream, lane-per-pixel sweep: retired in round 4) */
       CVS_FIR_KERNEL_TWO_PASS = 7,    /* two k_fir launches through an f32 frame (cached tables) */
       CVS_FIR_KERNEL_PASS = 8,        /* k_fir: one pass of the triangle scaler (both passes: two of these) */
       CVS_FIR_KERNEL_HV = 9,          /* k_fir_hv: per-line tables, horizontal pass first, gather per target line */
       CVS_FIR_KERNEL_WINDOW_PAIR = 10,    /* k_blur_pair: the register-window blur with two columns per lane (f16, up to 13 taps) */
       CVS_FIR_KERNEL_HALVE_PAIR = 11,     /* k_blur_halve_pair: blur + Lanczos halving with two source columns per lane (f16) */
       CVS_FIR_KERNEL_TILE_VH = 12 };      /* k_fir_tile_vh: the triangle scaler, vertical pass first, a workgroup per 128 x 16 tile (enlarging) */
CVS_EXPORT int cvs_fir_last_kernel(void);
/* How many FIR launches of the calling thread were chosen for a fused kernel that then did not launch and went to the next
 * kernel in line (same pixels, slower): each is also reported to the log handler as a warning.  A successful call leaves
 * cvs_last_error() empty either way. */
CVS_EXPORT int cvs_fir_fell_through_count(void);

/* ------------------------------------------------------------------ (3) fused chain: BASELINE config 2
 * out = f16( over-stack_{k=0..n-1}( f32( colour(layer_k) ) ) ), i.e. what the reference computes with
 * color.c on every layer, workspace.c:530-544 over the layers bottom-to-top and main.c:43-71 on the
 * result -- in ONE kernel: 8 B read per layer pixel, 8 B written per output pixel. */
#define CVS_CHAIN_MAX_LAYERS 8
typedef struct {
    rgba_frame_f16 *out;                               /* device frame; current_window is set */
    const rgba_frame_f16 *layers[CVS_CHAIN_MAX_LAYERS];/* device frames, bottom first */
    int nlayers;
} cvs_chain_job;
/* m == NULL: no colour stage -- the plain workspace stack of f16 layers (both tables must then be CVS_LUT_NONE).
 * Jobs are carried out as if one after the other: a job may read or overwrite what an earlier job of the same call
 * wrote (the library starts a new launch there); `out` may be one of the job's own layers (in place), but must not
 * overlap a layer at a shifted address (such a job takes the node-by-node path). */
CVS_EXPORT int cvs_chain_color_over_f16_dev(const cvs_chain_job *jobs, int njobs, const float m[9],
                                            int pre_lut, int post_lut, cvs_stream_t s);
/* how the last chain call ran: 1 = single fused kernel, 0 = node-by-node device kernels
 * (windows did not all cover the output's full window) */
CVS_EXPORT int cvs_chain_last_was_fused(void);
/* 1 when the calling thread's last scaler call (cvs_scale_bilinear_*_dev, video_scale_bilinear_f32) ran both passes in one
 * launch (sweep_vh_ops.hip when the vertical pass comes first, sweep_ops.hip otherwise; factors >= ~0.3); tests and tools */
CVS_EXPORT int cvs_scale_last_was_fused(void);
/* kernel launches the calling thread's last fused chain call was cut into (about eight 4K frames' worth of bytes each) */
CVS_EXPORT int cvs_chain_last_launch_count(void);
/* crossfade of two f16 frames, f16 result: widen, video_mix_cross_f32 (video_mix.c:107-235), truncate -- one launch when
 * every window is the whole output frame */
CVS_EXPORT int cvs_mix_cross_f16_dev(rgba_frame_f16 *out, const rgba_frame_f16 *a, const rgba_frame_f16 *b, float mix_b, cvs_stream_t stream);

#if defined(__cplusplus)
}
#endif
#endif
