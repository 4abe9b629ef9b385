/*
 * pyframework.h -- C helpers a CPython extension uses to plug into fluggo.media.process
 * (video path).  Source-compatible with the video and frame-function parts of the reference's
 * include/pyframework.h:24-93; the audio / clock / codec holders are not part of this library.
 *
 * Plugin protocol (pyframework.h:52, framework.h:185-194):
 *   a video source is any Python object with an attribute `_video_frame_source_funcs` holding a
 *   PyCapsule NAMED "_video_frame_source_funcs" around a static video_frame_source_funcs; the
 *   functions receive the PyObject* as `self`.  Slot 3 may carry a device-frame entry
 *   (flags & VIDEO_SOURCE_FLAG_DEVICE), see canvas_hip.h.
 *   a frame function is a constant (number, tuple, box2i, ...) or an object with a capsule
 *   `_frame_function_funcs` around a static FrameFunctionFuncs.
 */
#ifndef fluggo_pyframework
#define fluggo_pyframework

#include <Python.h>
#include "framework.h"      /* as pyframework.h:25: EXPORT, glib, then canvas_hip.h */

#if defined(__cplusplus)
extern "C" {
#endif

/* value conversions against fluggo.media.basetypes (src/process/basetypes.c:26-115) */
CVS_EXPORT bool py_parse_rational(PyObject *in, rational *out);
CVS_EXPORT PyObject *py_make_rational(rational *in);
CVS_EXPORT PyObject *py_make_rgba_f32(rgba_f32 *color);
CVS_EXPORT PyObject *py_make_box2f(box2f *box);
CVS_EXPORT PyObject *py_make_box2i(box2i *box);
CVS_EXPORT PyObject *py_make_v2f(v2f *v);
CVS_EXPORT PyObject *py_make_v2i(v2i *v);
CVS_EXPORT bool py_parse_rgba_f32(PyObject *obj, rgba_f32 *color);
CVS_EXPORT bool py_parse_box2f(PyObject *obj, box2f *box);
CVS_EXPORT bool py_parse_box2i(PyObject *obj, box2i *box);
CVS_EXPORT bool py_parse_v2f(PyObject *obj, v2f *v);
CVS_EXPORT bool py_parse_v2i(PyObject *obj, v2i *v);

/* video sources (src/process/main.c:30-66): *source is released first; NULL/None clears it */
#define VIDEO_FRAME_SOURCE_FUNCS "_video_frame_source_funcs"
CVS_EXPORT bool py_video_take_source(PyObject *obj, video_source **source);
CVS_EXPORT extern PyTypeObject py_type_VideoSource;

/* frame functions (pyframework.h:69-93, src/process/basicframefuncs.c:179-359) */
#define FRAME_FUNCTION_FUNCS "_frame_function_funcs"
typedef void (*framefunc_get_values_func)(PyObject *self, ssize_t count, double *frames, double (*out_values)[4]);
typedef struct { int flags; framefunc_get_values_func get_values; } FrameFunctionFuncs;
typedef struct {
    PyObject *source;
    PyObject *csource;
    FrameFunctionFuncs *funcs;
    double constant[4];
} FrameFunctionHolder;

CVS_EXPORT bool py_framefunc_take_source(PyObject *source, FrameFunctionHolder *holder);
CVS_EXPORT int framefunc_get_i32(FrameFunctionHolder *holder, double frame);
CVS_EXPORT float framefunc_get_f32(FrameFunctionHolder *holder, double frame);
CVS_EXPORT void framefunc_get_v2f(v2f *result, FrameFunctionHolder *holder, double frame);
CVS_EXPORT void framefunc_get_box2i(box2i *result, FrameFunctionHolder *holder, double frame);
CVS_EXPORT void framefunc_get_rgba_f32(rgba_f32 *result, FrameFunctionHolder *holder, double frame);
CVS_EXPORT void framefunc_init(FrameFunctionHolder *holder, double c0, double c1, double c2, double c3);
CVS_EXPORT extern PyTypeObject py_type_FrameFunction;
CVS_EXPORT extern PyTypeObject py_type_AnimationFunc;      /* src/process/AnimationFunc.c:349 */

/* coded images (pyframework.h:121-132, framework.h:510-523) */
#define CODED_IMAGE_SOURCE_FUNCS "_coded_image_source_funcs"
typedef coded_image *(*coded_image_getFrameFunc)(void *self, int frame, int quality_hint);
typedef struct { int flags; coded_image_getFrameFunc getFrame; } coded_image_source_funcs;
typedef struct { void *obj; coded_image_source_funcs *funcs; } coded_image_source;     /* obj: the PyObject, a strong reference */
typedef struct { coded_image_source source; PyObject *csource; } CodedImageSourceHolder;
CVS_EXPORT bool py_coded_image_take_source(PyObject *source, CodedImageSourceHolder *holder);
CVS_EXPORT extern PyTypeObject py_type_CodedImageSource;

#if defined(__cplusplus)
}
#endif
#endif
