/*
 * pymodule.c -- fluggo.media.process: module init, value conversions, the capsule protocol,
 * frame-function holders, node plumbing, module-level functions.
 *
 * Python-visible surface kept from the reference:
 *   src/process/main.c:30-66      py_video_take_source (capsule "_video_frame_source_funcs")
 *   src/process/main.c:120-167    time_get_frame (default window 4096x4096)
 *   src/process/main.c:252-268    get_frame_time, get_time_frame, enable_glib_logging, GL helpers
 *   src/process/basetypes.c       py_parse_* / py_make_* against fluggo.media.basetypes
 *   src/process/basicframefuncs.c:179-359   frame-function holders
 *   src/process/VideoSource.c     VideoSource base type
 * Everything is re-authored against libcanvas_hip.so; pixel work happens on the GPU only.
 */
#include "pyext.h"
#include <math.h>

/* ---------------------------------------------------------------- basetypes */

static PyObject *t_v2i, *t_v2f, *t_box2i, *t_box2f, *t_rgba, *t_fraction;

static int import_basetypes(void) {
    PyObject *m = PyImport_ImportModule("fluggo.media.basetypes");
    if (!m) return -1;
    t_v2i = PyObject_GetAttrString(m, "v2i");
    t_v2f = PyObject_GetAttrString(m, "v2f");
    t_box2i = PyObject_GetAttrString(m, "box2i");
    t_box2f = PyObject_GetAttrString(m, "box2f");
    t_rgba = PyObject_GetAttrString(m, "rgba");
    Py_DECREF(m);
    PyObject *f = PyImport_ImportModule("fractions");
    if (f) { t_fraction = PyObject_GetAttrString(f, "Fraction"); Py_DECREF(f); }
    return (t_v2i && t_v2f && t_box2i && t_box2f && t_rgba && t_fraction) ? 0 : -1;
}

CVS_EXPORT PyObject *py_make_v2i(v2i *v) { return PyObject_CallFunction(t_v2i, "ii", v->x, v->y); }
CVS_EXPORT PyObject *py_make_v2f(v2f *v) { return PyObject_CallFunction(t_v2f, "ff", v->x, v->y); }
CVS_EXPORT PyObject *py_make_box2i(box2i *b) { return PyObject_CallFunction(t_box2i, "(ii)(ii)", b->min.x, b->min.y, b->max.x, b->max.y); }
CVS_EXPORT PyObject *py_make_box2f(box2f *b) { return PyObject_CallFunction(t_box2f, "(ff)(ff)", b->min.x, b->min.y, b->max.x, b->max.y); }
CVS_EXPORT PyObject *py_make_rgba_f32(rgba_f32 *c) { return PyObject_CallFunction(t_rgba, "ffff", c->r, c->g, c->b, c->a); }
CVS_EXPORT PyObject *py_make_rational(rational *r) { return PyObject_CallFunction(t_fraction, "iI", r->n, r->d); }

CVS_EXPORT bool py_parse_v2i(PyObject *o, v2i *v) { return PyArg_ParseTuple(o, "ii", &v->x, &v->y) != 0; }
CVS_EXPORT bool py_parse_v2f(PyObject *o, v2f *v) { return PyArg_ParseTuple(o, "ff", &v->x, &v->y) != 0; }
CVS_EXPORT bool py_parse_box2i(PyObject *o, box2i *b) { return PyArg_ParseTuple(o, "(ii)(ii)", &b->min.x, &b->min.y, &b->max.x, &b->max.y) != 0; }
CVS_EXPORT bool py_parse_box2f(PyObject *o, box2f *b) { return PyArg_ParseTuple(o, "(ff)(ff)", &b->min.x, &b->min.y, &b->max.x, &b->max.y) != 0; }
CVS_EXPORT bool py_parse_rgba_f32(PyObject *o, rgba_f32 *c) { return PyArg_ParseTuple(o, "ffff", &c->r, &c->g, &c->b, &c->a) != 0; }

CVS_EXPORT bool py_parse_rational(PyObject *in, rational *out) {
    if (PyLong_Check(in)) { out->n = (int32_t)PyLong_AsLong(in); out->d = 1; return !PyErr_Occurred(); }
    PyObject *n = PyObject_GetAttrString(in, "numerator"), *d = n ? PyObject_GetAttrString(in, "denominator") : NULL;
    bool ok = n && d;
    if (ok) {
        out->n = (int32_t)PyLong_AsLong(n);
        out->d = (uint32_t)PyLong_AsUnsignedLong(d);
        ok = !PyErr_Occurred();
    }
    Py_XDECREF(n); Py_XDECREF(d);
    return ok;
}

/* ---------------------------------------------------------------- capsule protocol */

typedef struct { video_source source; PyObject *capsule; } held_source;

CVS_EXPORT bool py_video_take_source(PyObject *obj, video_source **source) {
    held_source *h = (held_source *)*source;
    if (h) {
        Py_CLEAR(h->source.obj);
        Py_CLEAR(h->capsule);
        PyMem_RawFree(h);
        *source = NULL;
    }
    if (!obj || obj == Py_None) return true;

    PyObject *capsule = PyObject_GetAttrString(obj, VIDEO_FRAME_SOURCE_FUNCS);
    if (!capsule || !PyCapsule_IsValid(capsule, VIDEO_FRAME_SOURCE_FUNCS)) {
        Py_XDECREF(capsule);
        PyErr_SetString(PyExc_Exception, "The source didn't have an acceptable " VIDEO_FRAME_SOURCE_FUNCS " attribute.");
        return false;
    }
    h = PyMem_RawMalloc(sizeof *h);
    if (!h) { Py_DECREF(capsule); PyErr_NoMemory(); return false; }
    Py_INCREF(obj);
    h->source.obj = obj;
    h->source.funcs = PyCapsule_GetPointer(capsule, VIDEO_FRAME_SOURCE_FUNCS);
    h->capsule = capsule;
    *source = &h->source;
    return true;
}

PyObject *pyext_capsule_getter(PyObject *self, void *closure) {
    PyObject *c = *(PyObject **)closure;
    Py_INCREF(c);
    return c;
}

int pyext_make_capsule(PyObject **slot, video_frame_source_funcs *funcs) {
    *slot = PyCapsule_New(funcs, VIDEO_FRAME_SOURCE_FUNCS, NULL);
    return *slot ? 0 : -1;
}

int pyext_add_type(PyObject *module, const char *name, PyTypeObject *type) {
    if (PyType_Ready(type) < 0) return -1;
    Py_INCREF(type);
    return PyModule_AddObject(module, name, (PyObject *)type);
}

/* ---------------------------------------------------------------- frame-function holders */

CVS_EXPORT void framefunc_init(FrameFunctionHolder *h, double c0, double c1, double c2, double c3) {
    h->source = NULL; h->csource = NULL; h->funcs = NULL;
    h->constant[0] = c0; h->constant[1] = c1; h->constant[2] = c2; h->constant[3] = c3;
}

CVS_EXPORT bool py_framefunc_take_source(PyObject *source, FrameFunctionHolder *h) {
    Py_CLEAR(h->source);
    Py_CLEAR(h->csource);
    memset(h, 0, sizeof *h);
    if (!source || source == Py_None) return true;

    /* constants first: box2i, box2f, 1..4-tuple, plain number (basicframefuncs.c:191-262) */
    box2i bi; box2f bf;
    if (PyTuple_Check(source) && PyTuple_GET_SIZE(source) == 2 && PyTuple_Check(PyTuple_GET_ITEM(source, 0))) {
        if (py_parse_box2i(source, &bi)) { h->constant[0] = bi.min.x; h->constant[1] = bi.min.y; h->constant[2] = bi.max.x; h->constant[3] = bi.max.y; return true; }
        PyErr_Clear();
        if (py_parse_box2f(source, &bf)) { h->constant[0] = bf.min.x; h->constant[1] = bf.min.y; h->constant[2] = bf.max.x; h->constant[3] = bf.max.y; return true; }
        PyErr_Clear();
    }
    if (PyTuple_Check(source)) {
        Py_ssize_t n = PyTuple_GET_SIZE(source);
        if (n == 0) { PyErr_SetString(PyExc_ValueError, "An empty tuple was passed."); return false; }
        if (n > 4) { PyErr_Format(PyExc_ValueError, "One of the tuples passed has more than four entries (%zd).", n); return false; }
        for (Py_ssize_t i = 0; i < n; i++) {
            PyObject *f = PyNumber_Float(PyTuple_GET_ITEM(source, i));
            if (!f) return false;
            h->constant[i] = PyFloat_AS_DOUBLE(f);
            Py_DECREF(f);
        }
        return true;
    }
    PyObject *as_float = PyNumber_Float(source);
    if (as_float) { h->constant[0] = PyFloat_AS_DOUBLE(as_float); Py_DECREF(as_float); return true; }
    PyErr_Clear();

    PyObject *capsule = PyObject_GetAttrString(source, FRAME_FUNCTION_FUNCS);
    if (!capsule || !PyCapsule_IsValid(capsule, FRAME_FUNCTION_FUNCS)) {
        Py_XDECREF(capsule);
        PyErr_SetString(PyExc_Exception, "The source didn't have an acceptable " FRAME_FUNCTION_FUNCS " attribute.");
        return false;
    }
    Py_INCREF(source);
    h->source = source;
    h->csource = capsule;
    h->funcs = PyCapsule_GetPointer(capsule, FRAME_FUNCTION_FUNCS);
    return true;
}

static void holder_eval(FrameFunctionHolder *h, double frame, double out[4]) {
    if (h->funcs && h->funcs->get_values) {
        double r[1][4];
        h->funcs->get_values(h->source, 1, &frame, r);
        memcpy(out, r[0], sizeof r[0]);
    } else {
        memcpy(out, h->constant, sizeof h->constant);
    }
}

CVS_EXPORT int framefunc_get_i32(FrameFunctionHolder *h, double frame) { double v[4]; holder_eval(h, frame, v); return (int)lround(v[0]); }
CVS_EXPORT float framefunc_get_f32(FrameFunctionHolder *h, double frame) { double v[4]; holder_eval(h, frame, v); return (float)v[0]; }
CVS_EXPORT void framefunc_get_v2f(v2f *r, FrameFunctionHolder *h, double frame) { double v[4]; holder_eval(h, frame, v); r->x = (float)v[0]; r->y = (float)v[1]; }
CVS_EXPORT void framefunc_get_box2i(box2i *r, FrameFunctionHolder *h, double frame) {
    double v[4];
    holder_eval(h, frame, v);
    /* lround of +-2^31 would overflow the int32 fields: the "everywhere" default window uses INT_MIN/INT_MAX */
    for (int i = 0; i < 4; i++) v[i] = v[i] < -2147483648.0 ? -2147483648.0 : (v[i] > 2147483647.0 ? 2147483647.0 : v[i]);
    box2i_set(r, (int)lround(v[0]), (int)lround(v[1]), (int)lround(v[2]), (int)lround(v[3]));
}
CVS_EXPORT void framefunc_get_rgba_f32(rgba_f32 *r, FrameFunctionHolder *h, double frame) {
    double v[4];
    holder_eval(h, frame, v);
    r->r = (float)v[0]; r->g = (float)v[1]; r->b = (float)v[2];
    r->a = clampf((float)v[3], 0.0f, 1.0f);              /* basicframefuncs.c:334-348 */
}

/* ---------------------------------------------------------------- node plumbing */

size_t frame_bytes(const box2i *full, int format) {
    v2i s;
    box2i_get_size(full, &s);
    return (size_t)s.x * (size_t)s.y * (format == CVS_FORMAT_F32 ? sizeof(rgba_f32) : sizeof(rgba_f16));
}

void node_get_frame_dev(PyObject *self, int frame_index, rgba_frame_dev *frame, int native, node_render_func render) {
    if (frame->format == native) { render(self, frame_index, frame); return; }
    /* other format requested: render natively into scratch with the same full window, then convert the
     * rows of current_window -- what video_get_frame_f16/f32 do on the host (main.c:43-71, 115-139) */
    rgba_frame_dev tmp = { NULL, native, frame->full_window, frame->full_window, frame->stream };
    tmp.data = cvs_pool_malloc(frame_bytes(&tmp.full_window, native), frame->stream);
    if (!tmp.data) { box2i_set_empty(&frame->current_window); return; }
    render(self, frame_index, &tmp);
    int rc = 0;
    if (!box2i_is_empty(&tmp.current_window)) {
        if (native == CVS_FORMAT_F16) {
            rgba_frame_f16 in = { tmp.data, tmp.full_window, tmp.current_window };
            rgba_frame_f32 out = { frame->data, frame->full_window, frame->full_window };
            rc = cvs_frame_f16_to_f32_dev(&out, &in, frame->stream);
        } else {
            rgba_frame_f32 in = { tmp.data, tmp.full_window, tmp.current_window };
            rgba_frame_f16 out = { frame->data, frame->full_window, frame->full_window };
            rc = cvs_frame_f32_to_f16_dev(&out, &in, frame->stream);
        }
    }
    frame->current_window = tmp.current_window;
    if (rc != 0) box2i_set_empty(&frame->current_window);
    cvs_pool_free(tmp.data, frame->stream);
}

static void host_edge(PyObject *self, int frame_index, void *host_data, const box2i *full, box2i *current,
                      int format, int native, node_render_func render) {
    rgba_frame_dev d = { NULL, format, *full, *full, NULL };
    size_t bytes = frame_bytes(full, format);
    d.data = cvs_pool_malloc(bytes, NULL);
    if (!d.data) { box2i_set_empty(current); return; }
    /* pixels outside current_window are undefined (docs/sphinx/cprocess/video.rst:44-47): nothing is
     * uploaded, one copy back */
    node_get_frame_dev(self, frame_index, &d, native, render);
    if (!box2i_is_empty(&d.current_window) && cvs_memcpy_d2h(host_data, d.data, bytes, NULL) != 0) box2i_set_empty(&d.current_window);
    *current = d.current_window;
    cvs_pool_free(d.data, NULL);
}

void node_get_frame_host16(PyObject *self, int frame_index, rgba_frame_f16 *frame, int native, node_render_func render) {
    host_edge(self, frame_index, frame->data, &frame->full_window, &frame->current_window, CVS_FORMAT_F16, native, render);
}

void node_get_frame_host32(PyObject *self, int frame_index, rgba_frame_f32 *frame, int native, node_render_func render) {
    host_edge(self, frame_index, frame->data, &frame->full_window, &frame->current_window, CVS_FORMAT_F32, native, render);
}

/* ---------------------------------------------------------------- base types */

static PyMethodDef VideoSource_methods[] = {
    { "get_frame_f16", (PyCFunction)py_get_frame_f16, METH_VARARGS | METH_KEYWORDS,
      "(RgbaFrameF16) frame = source.get_frame_f16(frame_index, data_window[, force_gl])" },
    { "get_frame_f32", (PyCFunction)py_get_frame_f32, METH_VARARGS | METH_KEYWORDS,
      "(RgbaFrameF32) frame = source.get_frame_f32(frame_index, data_window[, force_gl])" },
    { "get_frame_argb32", (PyCFunction)py_get_frame_argb32, METH_VARARGS | METH_KEYWORDS,
      "(bytearray or None, current_window) = source.get_frame_argb32(frame_index, data_window): premultiplied ARGB32 of the "
      "defined window, converted on the device (what get_frame_f16(...).to_argb32_bytes() returns, for half the download)" },
    { "get_frame_rgba8", (PyCFunction)py_get_frame_rgba8, METH_VARARGS | METH_KEYWORDS,
      "(bytearray or None, current_window) = source.get_frame_rgba8(frame_index, data_window, rendering_intent=1.25): "
      "sRGB-encoded r,g,b,a bytes of the defined window, converted on the device (the software widget's display conversion)" },
    { NULL }
};

CVS_EXPORT PyTypeObject py_type_VideoSource = {
    PyVarObject_HEAD_INIT(NULL, 0)
    .tp_name = "fluggo.media.process.VideoSource",
    .tp_basicsize = sizeof(PyObject),
    .tp_flags = Py_TPFLAGS_DEFAULT | Py_TPFLAGS_BASETYPE,
    .tp_methods = VideoSource_methods,
    .tp_new = PyType_GenericNew,
};

/* ---------------------------------------------------------------- module functions */

static PyObject *mod_get_frame_time(PyObject *self, PyObject *args) {
    PyObject *rate_obj; int frame; rational rate;
    if (!PyArg_ParseTuple(args, "Oi", &rate_obj, &frame)) return NULL;
    if (!py_parse_rational(rate_obj, &rate)) return NULL;
    return PyLong_FromLongLong(get_frame_time(&rate, frame));
}

static PyObject *mod_get_time_frame(PyObject *self, PyObject *args) {
    PyObject *rate_obj; long long time; rational rate;
    if (!PyArg_ParseTuple(args, "OL", &rate_obj, &time)) return NULL;
    if (!py_parse_rational(rate_obj, &rate)) return NULL;
    return PyLong_FromLong(get_time_frame(&rate, time));
}

/* src/process/main.c:120-167: pull min_frame..max_frame into one f16 frame, return elapsed ns */
static PyObject *mod_time_get_frame(PyObject *self, PyObject *args, PyObject *kw) {
    static char *kwlist[] = { "source", "min_frame", "max_frame", "data_window", NULL };
    PyObject *window_obj = NULL, *source_obj;
    int min_frame, max_frame;
    rgba_frame_f16 frame;
    box2i_set(&frame.full_window, 0, 0, 4095, 4095);
    if (!PyArg_ParseTupleAndKeywords(args, kw, "Oii|O", kwlist, &source_obj, &min_frame, &max_frame, &window_obj)) return NULL;
    if (window_obj && window_obj != Py_None) {
        if (!PyArg_ParseTuple(window_obj, "iiii", &frame.full_window.min.x, &frame.full_window.min.y, &frame.full_window.max.x, &frame.full_window.max.y)) {
            PyErr_Clear();
            if (!py_parse_box2i(window_obj, &frame.full_window)) return NULL;
        }
    }
    frame.current_window = frame.full_window;
    size_t bytes = frame_bytes(&frame.full_window, CVS_FORMAT_F16);
    frame.data = PyMem_Malloc(bytes ? bytes : 1);
    if (!frame.data) return PyErr_NoMemory();
    video_source *source = NULL;
    if (!py_video_take_source(source_obj, &source)) { PyMem_Free(frame.data); return NULL; }
    int64_t t0 = gettime();
    for (int i = min_frame; i <= max_frame; i++) video_get_frame_f16(source, i, &frame);
    int64_t t1 = gettime();
    PyMem_Free(frame.data);
    py_video_take_source(NULL, &source);
    return PyLong_FromLongLong(t1 - t0);
}

/* enable_glib_logging(enable): the reference bridges its glib log domains to Python's `logging` (main.c:171-191,
 * 272-329).  The library's diagnostics take the same road here: logging.getLogger(domain).warning(message), from
 * whichever thread the failure happened on (the handler takes the GIL itself). */
static PyObject *g_get_logger;

static void python_log_handler(const char *domain, int level, const char *message, void *user_data) {
    if (!Py_IsInitialized()) { fprintf(stderr, "%s: %s\n", domain, message); return; }
    PyGILState_STATE g = PyGILState_Ensure();
    PyObject *type, *value, *tb;
    PyErr_Fetch(&type, &value, &tb);                      /* a failing call may already carry a Python error */
    PyObject *logger = g_get_logger ? PyObject_CallFunction(g_get_logger, "s", domain) : NULL;
    PyObject *r = logger ? PyObject_CallMethod(logger, level == CVS_LOG_ERROR ? "error" : level == CVS_LOG_INFO ? "info" : "warning", "s", message) : NULL;
    if (!r) { PyErr_Clear(); fprintf(stderr, "%s: %s\n", domain, message); }
    Py_XDECREF(r);
    Py_XDECREF(logger);
    PyErr_Restore(type, value, tb);
    PyGILState_Release(g);
}

static PyObject *mod_enable_logging(PyObject *self, PyObject *args) {
    int enable = 1;
    if (!PyArg_ParseTuple(args, "|p", &enable)) return NULL;
    if (enable && !g_get_logger) {
        PyObject *logging = PyImport_ImportModule("logging");
        if (!logging) return NULL;
        g_get_logger = PyObject_GetAttrString(logging, "getLogger");
        Py_DECREF(logging);
        if (!g_get_logger) return NULL;
    }
    cvs_set_log_handler(enable ? python_log_handler : NULL, NULL);
    Py_RETURN_NONE;
}

/* GL helpers of the reference (main.c:193-250): callers exist in scripts/; the GL path is gone */
static PyObject *mod_gl_stub(PyObject *self, PyObject *args) { Py_RETURN_NONE; }
static PyObject *mod_check_context(PyObject *self, PyObject *args) { return PyBool_FromLong(cvs_device_count() > 0); }

static PyObject *mod_last_error(PyObject *self, PyObject *args) { return PyUnicode_FromString(cvs_last_error()); }
static PyObject *mod_device_name(PyObject *self, PyObject *args) { return PyUnicode_FromString(cvs_device_name()); }

/* set_arithmetic("separate" | "contracted") -> previous; get_arithmetic(): which of the reference's two builds the pixels
 * follow -- gcc -std=c99 (every multiply and add rounded on its own; the default) or clang (a * b + c inside an expression
 * fused, SConstruct:46-48,75-83).  Added beside the reference's surface (canvas_hip.h cvs_set_arithmetic). */
static PyObject *arith_name(int mode) { return PyUnicode_FromString(mode == CVS_ARITH_CONTRACTED ? "contracted" : "separate"); }
static PyObject *mod_get_arithmetic(PyObject *self, PyObject *args) { return arith_name(cvs_get_arithmetic()); }
static PyObject *mod_set_arithmetic(PyObject *self, PyObject *args) {
    const char *name;
    if (!PyArg_ParseTuple(args, "s", &name)) return NULL;
    int mode = strcmp(name, "separate") == 0 ? CVS_ARITH_SEPARATE : strcmp(name, "contracted") == 0 ? CVS_ARITH_CONTRACTED : -1;
    if (mode < 0) { PyErr_SetString(PyExc_ValueError, "arithmetic must be 'separate' or 'contracted'"); return NULL; }
    return arith_name(cvs_set_arithmetic(mode));
}

static PyObject *mod_device_count(PyObject *self, PyObject *args) { return PyLong_FromLong(cvs_device_count()); }
static PyObject *mod_frame_owner(PyObject *self, PyObject *args) {
    long long frame; int n;
    if (!PyArg_ParseTuple(args, "Li", &frame, &n)) return NULL;
    if (n <= 0) { PyErr_SetString(PyExc_ValueError, "frame_owner: the number of owners must be positive"); return NULL; }
    return PyLong_FromLong(cvs_frame_owner((int64_t)frame, n));
}

static PyMethodDef module_methods[] = {
    { "get_frame_time", mod_get_frame_time, METH_VARARGS, "get_frame_time(rate, frame) -> time in ns" },
    { "get_time_frame", mod_get_time_frame, METH_VARARGS, "get_time_frame(rate, time_ns) -> frame" },
    { "time_get_frame", (PyCFunction)mod_time_get_frame, METH_VARARGS | METH_KEYWORDS,
      "time_get_frame(source, min_frame, max_frame[, data_window=(0,0,4095,4095)]) -> elapsed ns" },
    { "enable_glib_logging", mod_enable_logging, METH_VARARGS, "enable_glib_logging(enable=True): route the library's diagnostics to logging.getLogger('fluggo.media.cprocess') instead of stderr." },
    { "create_offscreen_gl_context", mod_gl_stub, METH_VARARGS, "No-op: there is no GL path." },
    { "set_current_gl_context", mod_gl_stub, METH_VARARGS, "No-op: there is no GL path." },
    { "check_context_supported", mod_check_context, METH_VARARGS, "True when a HIP device is available." },
    { "set_arithmetic", mod_set_arithmetic, METH_VARARGS, "set_arithmetic('separate' | 'contracted') -> previous mode: follow the reference's gcc build (default) or its clang build (fused multiply-adds)." },
    { "get_arithmetic", mod_get_arithmetic, METH_NOARGS, "The arithmetic mode in force: 'separate' or 'contracted'." },
    { "device_count", mod_device_count, METH_NOARGS, "Number of HIP devices visible to this process (VideoPullQueue(devices=range(device_count())) uses them all)." },
    { "frame_owner", mod_frame_owner, METH_VARARGS, "frame_owner(frame_index, n) -> which of n devices / contexts / ranks renders the frame: frame_index mod n." },
    { "last_error", mod_last_error, METH_NOARGS, "Last error message of the calling thread." },
    { "device_name", mod_device_name, METH_NOARGS, "Name of the HIP device in use." },
    { NULL }
};

static struct PyModuleDef module_def = {
    PyModuleDef_HEAD_INIT, "process",
    "The Fluggo media processing library for Python, MI355X build: the per-pixel video path runs in HIP kernels.",
    -1, module_methods
};

PyMODINIT_FUNC PyInit_process(void) {
    PyObject *m = PyModule_Create(&module_def);
    if (!m) return NULL;
    if (import_basetypes() != 0) { Py_DECREF(m); return NULL; }
    init_half();
    if (pyext_add_type(m, "VideoSource", &py_type_VideoSource) < 0 || init_framefuncs(m) < 0 || init_animation(m) < 0 || init_frames(m) < 0 ||
        init_sources(m) < 0 || init_workspace(m) < 0 || init_dv(m) < 0) {
        Py_DECREF(m);
        return NULL;
    }
    return m;
}
