/*
 * pyframes.c -- RgbaFrameF16 / RgbaFrameF32 (host-resident result frames that are themselves video
 * sources), VideoSource.get_frame_f16/f32, and the constant-time frame functions.
 *
 * Surface kept from src/process/RgbaFrameF16.c:58-63,100-157,185-260 and RgbaFrameF32.c:
 *   attributes full_window / current_window (basetypes.box2i), len(frame), frame[i] -> rgba,
 *   pixel(x, y) -> rgba or None outside current_window, to_argb32_bytes(); a frame pulled through
 *   get_frame_f16/f32 re-windows itself with video_copy_frame_f16 / video_copy_frame_alpha_f32.
 * Frame functions: src/process/basicframefuncs.c:69-177 (LinearFrameFunc, LerpFunc) and
 *   :362-456 (FrameFunction.get_values).
 */
#include "pyext.h"

/* ---------------------------------------------------------------- frame objects */

typedef struct { PyObject_HEAD rgba_frame_f16 frame; } py_frame16;
typedef struct { PyObject_HEAD rgba_frame_f32 frame; } py_frame32;

static PyTypeObject py_type_RgbaFrameF16, py_type_RgbaFrameF32;
static PyObject *frame16_capsule, *frame32_capsule;

static void frame16_as_source16(py_frame16 *self, int frame_index, rgba_frame_f16 *out) { video_copy_frame_f16(out, &self->frame); }
static void frame32_as_source32(py_frame32 *self, int frame_index, rgba_frame_f32 *out) { video_copy_frame_alpha_f32(out, &self->frame, 1.0f); }

/* Slot 3 (the device slot) of a result frame: its host pixels re-windowed and uploaded -- video_get_frame_dev's own path for a
 * host-only source, reached through a vtable that has only the host slot.  It exists so that frame.get_frame_f32(...,
 * force_gl=True) yields pixels, as the reference's test expects (tests/process/video/SolidColorVideoSource.py:27-29), while
 * video_get_frame_f16_gl / _f32_gl keep the reference's rule for sources without slot 3 (an empty window, main.c:99-102). */
static video_frame_source_funcs frame16_host_funcs = { .flags = 0, .get_frame = (video_get_frame_func)frame16_as_source16 };
static video_frame_source_funcs frame32_host_funcs = { .flags = 0, .get_frame_32 = (video_get_frame_32_func)frame32_as_source32 };
static void frame16_as_source_dev(py_frame16 *self, int frame_index, rgba_frame_dev *out) {
    video_source host = { (void *)self, &frame16_host_funcs };
    video_get_frame_dev(&host, frame_index, out);
}
static void frame32_as_source_dev(py_frame32 *self, int frame_index, rgba_frame_dev *out) {
    video_source host = { (void *)self, &frame32_host_funcs };
    video_get_frame_dev(&host, frame_index, out);
}
static video_frame_source_funcs frame16_funcs = { .flags = VIDEO_SOURCE_FLAG_DEVICE, .get_frame = (video_get_frame_func)frame16_as_source16,
                                                  .get_frame_dev = (video_get_frame_dev_func)frame16_as_source_dev };
static video_frame_source_funcs frame32_funcs = { .flags = VIDEO_SOURCE_FLAG_DEVICE, .get_frame_32 = (video_get_frame_32_func)frame32_as_source32,
                                                  .get_frame_dev = (video_get_frame_dev_func)frame32_as_source_dev };

static void frame16_dealloc(py_frame16 *self) { PyMem_Free(self->frame.data); Py_TYPE(self)->tp_free((PyObject *)self); }
static void frame32_dealloc(py_frame32 *self) { PyMem_Free(self->frame.data); Py_TYPE(self)->tp_free((PyObject *)self); }

PyObject *py_RgbaFrameF16_new(box2i *full_window, rgba_frame_f16 **frame) {
    py_frame16 *f = (py_frame16 *)py_type_RgbaFrameF16.tp_alloc(&py_type_RgbaFrameF16, 0);
    if (!f) return NULL;
    f->frame.full_window = *full_window;
    box2i_set_empty(&f->frame.current_window);
    size_t bytes = frame_bytes(full_window, CVS_FORMAT_F16);
    f->frame.data = PyMem_Malloc(bytes ? bytes : 1);
    if (!f->frame.data) { Py_DECREF(f); return PyErr_NoMemory(); }
    if (frame) *frame = &f->frame;
    return (PyObject *)f;
}

PyObject *py_RgbaFrameF32_new(box2i *full_window, rgba_frame_f32 **frame) {
    py_frame32 *f = (py_frame32 *)py_type_RgbaFrameF32.tp_alloc(&py_type_RgbaFrameF32, 0);
    if (!f) return NULL;
    f->frame.full_window = *full_window;
    box2i_set_empty(&f->frame.current_window);
    size_t bytes = frame_bytes(full_window, CVS_FORMAT_F32);
    f->frame.data = PyMem_Malloc(bytes ? bytes : 1);
    if (!f->frame.data) { Py_DECREF(f); return PyErr_NoMemory(); }
    if (frame) *frame = &f->frame;
    return (PyObject *)f;
}

/* force_gl (RgbaFrameF16.c:221-249): "pull through vtable slot 3".  Slot 3 is the device slot here, so a true
 * value forces the source's device entry (video_get_frame_f16_gl / _f32_gl); there is no GL path. */
static bool parse_pull_args(PyObject *args, PyObject *kw, int *frame_index, box2i *window, bool *forced) {
    static char *kwlist[] = { "frame_index", "data_window", "force_gl", NULL };
    PyObject *window_obj = NULL, *force_gl = NULL;
    if (!PyArg_ParseTupleAndKeywords(args, kw, "iO|O", kwlist, frame_index, &window_obj, &force_gl)) return false;
    int truth = force_gl ? PyObject_IsTrue(force_gl) : 0;
    if (truth < 0) return false;
    *forced = truth != 0;
    return py_parse_box2i(window_obj, window);
}

PyObject *py_get_frame_f16(PyObject *self, PyObject *args, PyObject *kw) {
    int frame_index; box2i window; rgba_frame_f16 *frame; bool forced;
    if (!parse_pull_args(args, kw, &frame_index, &window, &forced)) return NULL;
    PyObject *result = py_RgbaFrameF16_new(&window, &frame);
    if (!result) return NULL;
    video_source *source = NULL;
    if (!py_video_take_source(self, &source)) { Py_DECREF(result); return NULL; }
    frame->current_window = frame->full_window;
    if (forced) video_get_frame_f16_gl(source, frame_index, frame);
    else video_get_frame_f16(source, frame_index, frame);
    py_video_take_source(NULL, &source);
    return result;
}

PyObject *py_get_frame_f32(PyObject *self, PyObject *args, PyObject *kw) {
    int frame_index; box2i window; rgba_frame_f32 *frame; bool forced;
    if (!parse_pull_args(args, kw, &frame_index, &window, &forced)) return NULL;
    PyObject *result = py_RgbaFrameF32_new(&window, &frame);
    if (!result) return NULL;
    video_source *source = NULL;
    if (!py_video_take_source(self, &source)) { Py_DECREF(result); return NULL; }
    frame->current_window = frame->full_window;
    if (forced) video_get_frame_f32_gl(source, frame_index, frame);
    else video_get_frame_f32(source, frame_index, frame);
    py_video_take_source(NULL, &source);
    return result;
}

/* Preview / thumbnail pulls: render on the device, convert to bytes on the device, and bring back 4 bytes per
 * pixel of the current window instead of the 8-byte halfs.  Returns (bytearray or None, current_window).
 * The two conversions are the reference's display edges:
 *   get_frame_argb32(frame_index, data_window): RgbaFrameF16.to_argb32_bytes (RgbaFrameF16.c:114-149);
 *   get_frame_rgba8(frame_index, data_window, rendering_intent=1.25): the software widget's rgba_u8
 *     (widget_gl.c:291-307: linear->sRGB table, then the ramp of widget_gl_set_rendering_intent, :955-968). */
static PyObject *pull_bytes(PyObject *self, PyObject *args, PyObject *kw, bool widget) {
    int frame_index; box2i window;
    float intent = 1.25f;                                /* widget_gl.c:428-429 */
    if (widget) {
        static char *kwlist[] = { "frame_index", "data_window", "rendering_intent", NULL };
        PyObject *window_obj = NULL;
        if (!PyArg_ParseTupleAndKeywords(args, kw, "iO|f", kwlist, &frame_index, &window_obj, &intent)) return NULL;
        if (!py_parse_box2i(window_obj, &window)) return NULL;
    } else if (!parse_pull_args(args, kw, &frame_index, &window, &(bool){ false })) return NULL;
    video_source *source = NULL;
    if (!py_video_take_source(self, &source)) return NULL;
    PyObject *bytes = NULL, *result = NULL;
    rgba_frame_dev d = { NULL, CVS_FORMAT_F16, window, window, NULL };
    void *packed = NULL;
    const size_t fbytes = frame_bytes(&window, CVS_FORMAT_F16);
    int rc = -1;
    if (!box2i_is_empty(&window)) {
        Py_BEGIN_ALLOW_THREADS
        d.data = cvs_pool_malloc(fbytes ? fbytes : 1, NULL);
        if (d.data) { video_get_frame_dev(source, frame_index, &d); rc = 0; }
        Py_END_ALLOW_THREADS
    } else { box2i_set_empty(&d.current_window); rc = 0; }
    if (rc == 0 && !box2i_is_empty(&d.current_window)) {
        v2i size;
        box2i_get_size(&d.current_window, &size);
        const size_t out = (size_t)size.x * (size_t)size.y * 4;
        bytes = PyByteArray_FromStringAndSize(NULL, (Py_ssize_t)out);
        if (bytes) {
            rgba_frame_f16 f = { d.data, d.full_window, d.current_window };
            char *host = PyByteArray_AS_STRING(bytes);
            Py_BEGIN_ALLOW_THREADS
            packed = cvs_pool_malloc(out, NULL);
            rc = !packed ? -1 : widget ? cvs_frame_to_rgba8_intent_dev(packed, &f, CVS_LUT_LINEAR_TO_SRGB, intent, NULL)
                                       : cvs_frame_to_bytes_dev(packed, &f, CVS_LUT_NONE, CVS_DISPLAY_ARGB32_PREMUL, NULL);
            if (rc == 0) rc = cvs_memcpy_d2h(host, packed, out, NULL);
            Py_END_ALLOW_THREADS
        } else rc = -2;
    }
    cvs_pool_free(packed, NULL);
    cvs_pool_free(d.data, NULL);
    py_video_take_source(NULL, &source);
    if (rc == -2) return NULL;                       /* Python error already set */
    if (rc != 0) { Py_XDECREF(bytes); PyErr_SetString(PyExc_RuntimeError, cvs_last_error()); return NULL; }
    PyObject *win = py_make_box2i(&d.current_window);
    if (win) result = Py_BuildValue("(ON)", bytes ? bytes : Py_None, win);
    Py_XDECREF(bytes);
    return result;
}

PyObject *py_get_frame_argb32(PyObject *self, PyObject *args, PyObject *kw) { return pull_bytes(self, args, kw, false); }
PyObject *py_get_frame_rgba8(PyObject *self, PyObject *args, PyObject *kw) { return pull_bytes(self, args, kw, true); }

static PyObject *frame16_full(py_frame16 *self, void *c) { return py_make_box2i(&self->frame.full_window); }
static PyObject *frame16_current(py_frame16 *self, void *c) { return py_make_box2i(&self->frame.current_window); }
static PyObject *frame32_full(py_frame32 *self, void *c) { return py_make_box2i(&self->frame.full_window); }
static PyObject *frame32_current(py_frame32 *self, void *c) { return py_make_box2i(&self->frame.current_window); }

static Py_ssize_t frame16_len(py_frame16 *self) { v2i s; box2i_get_size(&self->frame.full_window, &s); return (Py_ssize_t)s.x * s.y; }
static Py_ssize_t frame32_len(py_frame32 *self) { v2i s; box2i_get_size(&self->frame.full_window, &s); return (Py_ssize_t)s.x * s.y; }

/* one pixel f16 -> rgba: through the library's converter (half.c pointer), like RgbaFrameF16.c:74-79 */
static PyObject *pixel16_to_python(const rgba_f16 *p) {
    rgba_f32 c;
    half_convert_to_float(&c.r, &p->r, 4);
    return py_make_rgba_f32(&c);
}

static PyObject *frame16_item(py_frame16 *self, Py_ssize_t i) {
    if (i < 0 || i >= frame16_len(self)) { PyErr_SetString(PyExc_IndexError, "Index was out of range."); return NULL; }
    return pixel16_to_python(&self->frame.data[i]);
}
static PyObject *frame32_item(py_frame32 *self, Py_ssize_t i) {
    if (i < 0 || i >= frame32_len(self)) { PyErr_SetString(PyExc_IndexError, "Index was out of range."); return NULL; }
    return py_make_rgba_f32(&self->frame.data[i]);
}

static bool inside(const box2i *w, int x, int y) { return x >= w->min.x && x <= w->max.x && y >= w->min.y && y <= w->max.y; }

static PyObject *frame16_pixel(py_frame16 *self, PyObject *args) {
    int x, y;
    if (!PyArg_ParseTuple(args, "ii", &x, &y)) return NULL;
    if (!inside(&self->frame.current_window, x, y)) Py_RETURN_NONE;
    return pixel16_to_python(video_get_pixel_f16(&self->frame, x, y));
}
static PyObject *frame32_pixel(py_frame32 *self, PyObject *args) {
    int x, y;
    if (!PyArg_ParseTuple(args, "ii", &x, &y)) return NULL;
    if (!inside(&self->frame.current_window, x, y)) Py_RETURN_NONE;
    return py_make_rgba_f32(video_get_pixel_f32(&self->frame, x, y));
}

/* RgbaFrameF16.c:114-149: gamma-0.45 ramp per channel, premultiplied ARGB32, current_window only */
static PyObject *frame16_to_argb32(py_frame16 *self, PyObject *args) {
    const box2i *w = &self->frame.current_window;
    if (box2i_is_empty(w)) Py_RETURN_NONE;
    v2i size;
    box2i_get_size(w, &size);
    PyObject *result = PyByteArray_FromStringAndSize(NULL, (Py_ssize_t)size.x * size.y * 4);
    if (!result) return NULL;
    /* ramp + premultiplied packing run on the device (RgbaFrameF16.c:114-149 semantics) */
    if (video_frame_to_bytes(PyByteArray_AS_STRING(result), &self->frame, CVS_LUT_NONE, CVS_DISPLAY_ARGB32_PREMUL) != 0) {
        Py_DECREF(result);
        PyErr_SetString(PyExc_RuntimeError, cvs_last_error());
        return NULL;
    }
    return result;
}

static PyGetSetDef frame16_getset[] = {
    { VIDEO_FRAME_SOURCE_FUNCS, pyext_capsule_getter, NULL, "Video frame source C API.", &frame16_capsule },
    { "full_window", (getter)frame16_full, NULL, "The full data window for this frame." },
    { "current_window", (getter)frame16_current, NULL, "The current (defined) data window for this frame." },
    { NULL }
};
static PyGetSetDef frame32_getset[] = {
    { VIDEO_FRAME_SOURCE_FUNCS, pyext_capsule_getter, NULL, "Video frame source C API.", &frame32_capsule },
    { "full_window", (getter)frame32_full, NULL, "The full data window for this frame." },
    { "current_window", (getter)frame32_current, NULL, "The current (defined) data window for this frame." },
    { NULL }
};
static PyMethodDef frame16_methods[] = {
    { "pixel", (PyCFunction)frame16_pixel, METH_VARARGS, "pixel(x, y) -> rgba, or None outside current_window" },
    { "to_argb32_bytes", (PyCFunction)frame16_to_argb32, METH_VARARGS, "Premultiplied ARGB32 of the defined window (gamma 0.45), for QImage." },
    { NULL }
};
static PyMethodDef frame32_methods[] = {
    { "pixel", (PyCFunction)frame32_pixel, METH_VARARGS, "pixel(x, y) -> rgba, or None outside current_window" },
    { NULL }
};
static PySequenceMethods frame16_seq = { .sq_length = (lenfunc)frame16_len, .sq_item = (ssizeargfunc)frame16_item };
static PySequenceMethods frame32_seq = { .sq_length = (lenfunc)frame32_len, .sq_item = (ssizeargfunc)frame32_item };

static PyTypeObject py_type_RgbaFrameF16 = {
    PyVarObject_HEAD_INIT(NULL, 0)
    .tp_name = "fluggo.media.process.RgbaFrameF16", .tp_basicsize = sizeof(py_frame16), .tp_flags = Py_TPFLAGS_DEFAULT,
    .tp_base = &py_type_VideoSource, .tp_dealloc = (destructor)frame16_dealloc,
    .tp_getset = frame16_getset, .tp_methods = frame16_methods, .tp_as_sequence = &frame16_seq,
};
static PyTypeObject py_type_RgbaFrameF32 = {
    PyVarObject_HEAD_INIT(NULL, 0)
    .tp_name = "fluggo.media.process.RgbaFrameF32", .tp_basicsize = sizeof(py_frame32), .tp_flags = Py_TPFLAGS_DEFAULT,
    .tp_base = &py_type_VideoSource, .tp_dealloc = (destructor)frame32_dealloc,
    .tp_getset = frame32_getset, .tp_methods = frame32_methods, .tp_as_sequence = &frame32_seq,
};

int init_frames(PyObject *module) {
    if (pyext_make_capsule(&frame16_capsule, &frame16_funcs) < 0 || pyext_make_capsule(&frame32_capsule, &frame32_funcs) < 0) return -1;
    if (pyext_add_type(module, "RgbaFrameF16", &py_type_RgbaFrameF16) < 0) return -1;
    return pyext_add_type(module, "RgbaFrameF32", &py_type_RgbaFrameF32);
}

/* ---------------------------------------------------------------- frame functions */

static PyObject *framefunc_get_values(PyObject *self, PyObject *args) {
    PyObject *frames_obj;
    if (!PyArg_ParseTuple(args, "O", &frames_obj)) return NULL;
    Py_ssize_t count = 1;
    double *frames = NULL;
    if (PySequence_Check(frames_obj)) {
        PyObject *fast = PySequence_Fast(frames_obj, "expected a number or a sequence of numbers");
        if (!fast) return NULL;
        count = PySequence_Fast_GET_SIZE(fast);
        frames = PyMem_Malloc(sizeof(double) * (size_t)(count ? count : 1));
        for (Py_ssize_t i = 0; frames && i < count; i++) frames[i] = PyFloat_AsDouble(PySequence_Fast_GET_ITEM(fast, i));
        Py_DECREF(fast);
    } else {
        frames = PyMem_Malloc(sizeof(double));
        if (frames) frames[0] = PyFloat_AsDouble(frames_obj);
    }
    if (!frames) return PyErr_NoMemory();
    if (PyErr_Occurred()) { PyMem_Free(frames); return NULL; }

    FrameFunctionHolder holder = { 0 };
    if (!py_framefunc_take_source(self, &holder)) { PyMem_Free(frames); return NULL; }
    PyObject *result = PyList_New(count);
    double (*values)[4] = PyMem_Malloc(sizeof(double) * 4 * (size_t)(count ? count : 1));
    if (result && values) {
        if (holder.funcs && holder.funcs->get_values) holder.funcs->get_values(holder.source, count, frames, values);
        else for (Py_ssize_t i = 0; i < count; i++) memcpy(values[i], holder.constant, sizeof holder.constant);
        for (Py_ssize_t i = 0; i < count; i++)
            PyList_SET_ITEM(result, i, Py_BuildValue("dddd", values[i][0], values[i][1], values[i][2], values[i][3]));
    }
    PyMem_Free(values);
    PyMem_Free(frames);
    py_framefunc_take_source(NULL, &holder);
    return result;
}

static PyMethodDef FrameFunction_methods[] = {
    { "get_values", framefunc_get_values, METH_VARARGS, "value_list = func.get_values(frame or [frames])" },
    { NULL }
};

CVS_EXPORT PyTypeObject py_type_FrameFunction = {
    PyVarObject_HEAD_INIT(NULL, 0)
    .tp_name = "fluggo.media.process.FrameFunction", .tp_basicsize = sizeof(PyObject),
    .tp_flags = Py_TPFLAGS_DEFAULT | Py_TPFLAGS_BASETYPE, .tp_methods = FrameFunction_methods, .tp_new = PyType_GenericNew,
};

/* f(frame) = a * frame + b in slot 0 (basicframefuncs.c:69-101) */
typedef struct { PyObject_HEAD double a, b; } py_linear;
static PyObject *linear_capsule;
static int linear_init(py_linear *self, PyObject *args, PyObject *kw) {
    static char *kwlist[] = { "a", "b", NULL };
    return PyArg_ParseTupleAndKeywords(args, kw, "dd", kwlist, &self->a, &self->b) ? 0 : -1;
}
static void linear_values(py_linear *self, ssize_t count, double *frames, double (*out)[4]) {
    for (ssize_t i = 0; i < count; i++) { out[i][0] = frames[i] * self->a + self->b; out[i][1] = out[i][2] = out[i][3] = 0.0; }
}
static FrameFunctionFuncs linear_funcs = { 0, (framefunc_get_values_func)linear_values };
static PyGetSetDef linear_getset[] = { { FRAME_FUNCTION_FUNCS, pyext_capsule_getter, NULL, "Frame function C API.", &linear_capsule }, { NULL } };
static PyTypeObject py_type_LinearFrameFunc = {
    PyVarObject_HEAD_INIT(NULL, 0)
    .tp_name = "fluggo.media.process.LinearFrameFunc", .tp_basicsize = sizeof(py_linear), .tp_flags = Py_TPFLAGS_DEFAULT,
    .tp_base = &py_type_FrameFunction, .tp_new = PyType_GenericNew, .tp_init = (initproc)linear_init, .tp_getset = linear_getset,
};

/* four-slot linear interpolation start -> end over `length` frames, extrapolating (basicframefuncs.c:104-177) */
typedef struct { PyObject_HEAD float start[4], end[4]; double length; } py_lerp;
static PyObject *lerp_capsule;
static bool read4(PyObject *obj, float out[4], const char *what) {
    PyObject *fast = PySequence_Fast(obj, what);
    if (!fast) return false;
    for (Py_ssize_t i = 0; i < 4; i++)
        out[i] = i < PySequence_Fast_GET_SIZE(fast) ? (float)PyFloat_AsDouble(PySequence_Fast_GET_ITEM(fast, i)) : 0.0f;
    Py_DECREF(fast);
    return !PyErr_Occurred();
}
static int lerp_init(py_lerp *self, PyObject *args, PyObject *kw) {
    static char *kwlist[] = { "start", "end", "length", NULL };
    PyObject *s, *e;
    if (!PyArg_ParseTupleAndKeywords(args, kw, "OOd", kwlist, &s, &e, &self->length)) return -1;
    if (self->length <= 0.0) { PyErr_SetString(PyExc_Exception, "length must be greater than zero."); return -1; }
    /* box2i-style arguments ((x0,y0),(x1,y1)) flatten to four numbers */
    box2f b;
    if (PyTuple_Check(s) && PyTuple_GET_SIZE(s) == 2 && PyTuple_Check(PyTuple_GET_ITEM(s, 0)) && py_parse_box2f(s, &b)) { self->start[0] = b.min.x; self->start[1] = b.min.y; self->start[2] = b.max.x; self->start[3] = b.max.y; }
    else { PyErr_Clear(); if (!read4(s, self->start, "Expected a tuple or list for start.")) return -1; }
    if (PyTuple_Check(e) && PyTuple_GET_SIZE(e) == 2 && PyTuple_Check(PyTuple_GET_ITEM(e, 0)) && py_parse_box2f(e, &b)) { self->end[0] = b.min.x; self->end[1] = b.min.y; self->end[2] = b.max.x; self->end[3] = b.max.y; }
    else { PyErr_Clear(); if (!read4(e, self->end, "Expected a tuple or list for end.")) return -1; }
    return 0;
}
static void lerp_values(py_lerp *self, ssize_t count, double *frames, double (*out)[4]) {
    for (ssize_t i = 0; i < count; i++)
        for (int k = 0; k < 4; k++)
            out[i][k] = frames[i] * (self->end[k] - self->start[k]) / self->length + self->start[k];
}
static FrameFunctionFuncs lerp_funcs = { 0, (framefunc_get_values_func)lerp_values };
static PyGetSetDef lerp_getset[] = { { FRAME_FUNCTION_FUNCS, pyext_capsule_getter, NULL, "Frame function C API.", &lerp_capsule }, { NULL } };
static PyTypeObject py_type_LerpFunc = {
    PyVarObject_HEAD_INIT(NULL, 0)
    .tp_name = "fluggo.media.process.LerpFunc", .tp_basicsize = sizeof(py_lerp), .tp_flags = Py_TPFLAGS_DEFAULT,
    .tp_base = &py_type_FrameFunction, .tp_new = PyType_GenericNew, .tp_init = (initproc)lerp_init, .tp_getset = lerp_getset,
};

/* FrameFuncPassThroughFilter(source, offset=0.0): the upstream function evaluated at frame + offset; a constant
 * source passes through as that constant (src/process/FrameFuncPassThroughFilter.c:24-190).  Reader lock around the
 * upstream call, writer lock where the source is replaced (:60,121-130). */
typedef struct { PyObject_HEAD FrameFunctionHolder source; double offset; pthread_rwlock_t lock; bool ready; } py_ffpass;
static PyObject *ffpass_capsule;
static int ffpass_init(py_ffpass *self, PyObject *args, PyObject *kw) {
    static char *kwlist[] = { "source", "offset", NULL };
    PyObject *src;
    self->offset = 0.0;
    if (!PyArg_ParseTupleAndKeywords(args, kw, "O|d", kwlist, &src, &self->offset)) return -1;
    if (!self->ready) { pthread_rwlock_init(&self->lock, NULL); self->ready = true; }
    return py_framefunc_take_source(src, &self->source) ? 0 : -1;
}
static void ffpass_dealloc(py_ffpass *self) {
    py_framefunc_take_source(NULL, &self->source);
    if (self->ready) pthread_rwlock_destroy(&self->lock);
    Py_TYPE(self)->tp_free((PyObject *)self);
}
static void ffpass_values(py_ffpass *self, ssize_t count, double *frames, double (*out)[4]) {
    py_rdlock(&self->lock);
    if (self->source.funcs && self->source.funcs->get_values) {
        double *shifted = NULL;
        if (self->offset != 0.0 && count > 0) {
            shifted = malloc(sizeof(double) * (size_t)count);
            for (ssize_t i = 0; shifted && i < count; i++) shifted[i] = frames[i] + self->offset;
        }
        self->source.funcs->get_values(self->source.source, count, shifted ? shifted : frames, out);
        free(shifted);
    } else {
        for (ssize_t i = 0; i < count; i++) memcpy(out[i], self->source.constant, sizeof self->source.constant);
    }
    pthread_rwlock_unlock(&self->lock);
}
static FrameFunctionFuncs ffpass_funcs = { 0, (framefunc_get_values_func)ffpass_values };
static PyObject *ffpass_get_source(py_ffpass *self, PyObject *noargs) {
    PyObject *o = self->source.source ? self->source.source : Py_None;
    Py_INCREF(o);
    return o;
}
static PyObject *ffpass_set_source(py_ffpass *self, PyObject *args) {
    PyObject *src;
    if (!PyArg_ParseTuple(args, "O", &src)) return NULL;
    py_wrlock_nogil(&self->lock);
    const bool ok = py_framefunc_take_source(src, &self->source);
    pthread_rwlock_unlock(&self->lock);
    if (!ok) return NULL;
    Py_RETURN_NONE;
}
static PyObject *ffpass_get_offset(py_ffpass *self, void *c) { return PyFloat_FromDouble(self->offset); }
static int ffpass_set_offset(py_ffpass *self, PyObject *value, void *c) {
    const double v = value ? PyFloat_AsDouble(value) : 0.0;
    if (!value) { PyErr_SetString(PyExc_TypeError, "offset cannot be deleted"); return -1; }
    if (v == -1.0 && PyErr_Occurred()) return -1;
    self->offset = v;
    return 0;
}
static PyMethodDef ffpass_methods[] = {
    { "source", (PyCFunction)ffpass_get_source, METH_NOARGS, "Gets the source frame function." },
    { "set_source", (PyCFunction)ffpass_set_source, METH_VARARGS, "Sets the source frame function." },
    { NULL }
};
static PyGetSetDef ffpass_getset[] = {
    { FRAME_FUNCTION_FUNCS, pyext_capsule_getter, NULL, "Frame function C API.", &ffpass_capsule },
    { "offset", (getter)ffpass_get_offset, (setter)ffpass_set_offset, "Get or set the offset." },
    { NULL }
};
static PyTypeObject py_type_FrameFuncPassThroughFilter = {
    PyVarObject_HEAD_INIT(NULL, 0)
    .tp_name = "fluggo.media.process.FrameFuncPassThroughFilter", .tp_basicsize = sizeof(py_ffpass),
    .tp_flags = Py_TPFLAGS_DEFAULT | Py_TPFLAGS_BASETYPE, .tp_base = &py_type_FrameFunction, .tp_new = PyType_GenericNew,
    .tp_init = (initproc)ffpass_init, .tp_dealloc = (destructor)ffpass_dealloc, .tp_getset = ffpass_getset, .tp_methods = ffpass_methods,
};

int init_framefuncs(PyObject *module) {
    ffpass_capsule = PyCapsule_New(&ffpass_funcs, FRAME_FUNCTION_FUNCS, NULL);
    if (!ffpass_capsule || pyext_add_type(module, "FrameFunction", &py_type_FrameFunction) < 0) return -1;
    if (pyext_add_type(module, "FrameFuncPassThroughFilter", &py_type_FrameFuncPassThroughFilter) < 0) return -1;
    linear_capsule = PyCapsule_New(&linear_funcs, FRAME_FUNCTION_FUNCS, NULL);
    lerp_capsule = PyCapsule_New(&lerp_funcs, FRAME_FUNCTION_FUNCS, NULL);
    if (!linear_capsule || !lerp_capsule) return -1;
    if (pyext_add_type(module, "LinearFrameFunc", &py_type_LinearFrameFunc) < 0) return -1;
    return pyext_add_type(module, "LerpFunc", &py_type_LerpFunc);
}
