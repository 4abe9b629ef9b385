/*
 * internal.h -- shared by the host C files of libcanvas_hip.so.  Not installed.
 */
#ifndef CVS_INTERNAL_H
#define CVS_INTERNAL_H

#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "canvas_hip.h"
#include "../kernels/kernels.h"

/* ---- error plumbing: frame functions return void in the reference and signal failure through an
 * empty current_window (src/cprocess/main.c:35-38); the message goes to cvs_last_error() and, like
 * a g_warning, to stderr. */
void cvs_set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
void cvs_clear_error(void);
/* a notice for the log handler (or stderr) that is NOT an error of the call in progress: cvs_last_error() is left alone */
void cvs_log_warning(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

#define CVS_HIP(expr)                                                                          \
    do {                                                                                       \
        hipError_t cvs_e_ = (expr);                                                            \
        if (cvs_e_ != hipSuccess) {                                                            \
            cvs_set_error("%s: %s (%s:%d)", #expr, hipGetErrorString(cvs_e_), __FILE__, __LINE__); \
            return (int)cvs_e_ ? (int)cvs_e_ : -1;                                             \
        }                                                                                      \
    } while (0)

/* runtime.c: see there -- 1 when a graph being captured on `st` took over the hold on a cached table */
int cvs_capture_hold(hipStream_t st, void (*release)(void *), void *arg);

#define CVS_KERNEL(expr)                                                                       \
    do {                                                                                       \
        int cvs_k_ = (expr);                                                                   \
        if (cvs_k_ != 0) {                                                                     \
            cvs_set_error("%s: %s (%s:%d)", #expr, hipGetErrorString((hipError_t)cvs_k_), __FILE__, __LINE__); \
            return cvs_k_;                                                                     \
        }                                                                                      \
    } while (0)

/* The arithmetic flavour of the call the calling thread is in (canvas_hip.h cvs_set_arithmetic; snapshot taken by
 * cvs_enter()), and the launcher of that flavour: CVK(cvk_blur)(&bp, cus, s) is cvk_blur or cvk_blur_fma (kernels.h). */
int cvs_arith(void);
#define CVK(name) (cvs_arith() ? name##_fma : name)

/* Device contexts (runtime.c): the id of the context the calling thread's current call runs in -- set by cvs_enter().
 * Whatever a file keeps ON the device for the library's own use (cached tables) is kept per context: arrays of
 * CVS_MAX_CONTEXTS, indexed with cvs_ctx(). */
#define CVS_MAX_CONTEXTS 64
int cvs_ctx(void);

/* binds the calling thread to its context's device (cvs_set_context, else the default context; lazily opens context 0 on
 * CVS_DEVICE, default 0) and takes the snapshots of the call: context id, arithmetic flavour.  0 on success. */
int cvs_enter(void);
/* the stream to enqueue on: the caller's, or this thread's own when NULL */
hipStream_t cvs_pick_stream(cvs_stream_t s);
int cvs_cus(void);

static inline size_t cvs_box_pixels(const box2i *b) {
    v2i s;
    box2i_get_size(b, &s);
    return (size_t)s.x * (size_t)s.y;
}

static inline cvk_view cvs_view(void *data, const box2i *full) {
    cvk_view v;
    v.data = data;
    v.pitch = full->max.x < full->min.x ? 0 : full->max.x - full->min.x + 1;
    v.fx0 = full->min.x; v.fy0 = full->min.y; v.fx1 = full->max.x; v.fy1 = full->max.y;
    return v;
}

static inline cvk_rect cvs_rect(const box2i *b) {
    cvk_rect r = { b->min.x, b->min.y, b->max.x, b->max.y };
    return r;
}

static inline bool cvs_box_contains(const box2i *outer, const box2i *inner) {
    return box2i_is_empty(inner) ||
           (inner->min.x >= outer->min.x && inner->min.y >= outer->min.y && inner->max.x <= outer->max.x && inner->max.y <= outer->max.y);
}

/* A frame whose current_window reaches outside its own buffer would send a kernel out of bounds: refuse it here,
 * loudly, with the output marked empty.  (The reference would read past the buffer.) */
#define CVS_REQUIRE_INSIDE(in_frame, out_frame, what)                                                      \
    do {                                                                                                   \
        if (!cvs_box_contains(&(in_frame)->full_window, &(in_frame)->current_window)) {                   \
            cvs_set_error("%s: the input's current_window lies outside its buffer", what);                \
            box2i_set_empty(&(out_frame)->current_window);                                                 \
            return -1;                                                                                     \
        }                                                                                                  \
    } while (0)

/* ---- staging of host frames for the reference-named entry points (H2D -> kernels -> D2H) */
typedef struct {
    void *dev;            /* device copy of the whole full_window buffer (from the stream-ordered pool) */
    size_t bytes;
    hipStream_t stream;   /* the stream it was staged on: the block goes back to the pool on it */
} cvs_staged;

int cvs_stage_in(cvs_staged *st, const void *host, size_t bytes, int upload, hipStream_t s);
int cvs_stage_out(cvs_staged *st, void *host, hipStream_t s);      /* D2H + sync */
void cvs_stage_free(cvs_staged *st);

/* device LUT for an id, NULL for CVS_LUT_NONE; builds the tables on first use */
const half *cvs_lut_dev_or_null(int which);

#endif
