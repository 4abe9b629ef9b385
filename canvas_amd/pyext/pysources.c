/*
 * pysources.c -- the filter / source node types of fluggo.media.process.
 *
 * Constructor signatures, attributes and per-frame semantics follow the reference's Python-visible
 * types; the pixel work is a device render (see pyext.h):
 *   SolidColorVideoSource(color[, window])              src/process/SolidColorVideoSource.c:32-101
 *   EmptyVideoSource()                                  src/process/EmptyVideoSource.c
 *   VideoGainOffsetFilter(source, gain=1, offset=0)     src/process/VideoGainOffsetFilter.c:41-82,174-187
 *                                                       (GL-only there; formula video_filter.c:34-39)
 *   VideoMixFilter(src_a, src_b, mix_b)                 src/process/VideoMixFilter.c:41-66 (crossfade)
 *   VideoScaler(source, target_point, source_point, scale_factors, source_rect)   src/process/VideoScaler.c:38-135
 *   VideoPassThroughFilter(source, offset=0, start_frame=None, end_frame=None)    src/process/VideoPassThroughFilter.c:46-283
 *   VideoSequence() list of (source, offset, length)    src/process/VideoSequence.c:58-343
 *   Pulldown23RemovalFilter(source, offset)             src/process/Pulldown23RemovalFilter.c:31-107
 * Locking rule kept (VideoPassThroughFilter.c:121-148): reader lock around any upstream pull, writer
 * lock where a source reference is replaced.
 */
#include "pyext.h"
#include <limits.h>

#define UNUSED __attribute__((unused))

static void pull_dev(video_source *src, int frame_index, rgba_frame_dev *frame) { video_get_frame_dev(src, frame_index, frame); }

static rgba_frame_dev scratch_like(const rgba_frame_dev *f, int format, const box2i *full) {
    rgba_frame_dev t = { NULL, format, *full, *full, f->stream };
    t.data = cvs_pool_malloc(frame_bytes(full, format), f->stream);
    return t;
}

/* ---------------------------------------------------------------- SolidColorVideoSource */

typedef struct { PyObject_HEAD FrameFunctionHolder window, color; } py_solid;

static int solid_init(py_solid *self, PyObject *args, PyObject *kw) {
    PyObject *window_obj = NULL, *color_obj;
    if (!PyArg_ParseTuple(args, "O|O", &color_obj, &window_obj)) return -1;
    if (!py_framefunc_take_source(color_obj, &self->color)) return -1;
    framefunc_init(&self->window, INT_MIN, INT_MIN, INT_MAX, INT_MAX);       /* everywhere */
    if (window_obj && window_obj != Py_None && !py_framefunc_take_source(window_obj, &self->window)) return -1;
    return 0;
}
static void solid_dealloc(py_solid *self) {
    py_framefunc_take_source(NULL, &self->window);
    py_framefunc_take_source(NULL, &self->color);
    Py_TYPE(self)->tp_free((PyObject *)self);
}
/* renders in whichever format is asked for: both are native (SolidColorVideoSource.c fills both slots) */
static void solid_slot_dev(py_solid *self, int frame_index, rgba_frame_dev *f) {
    box2i window; rgba_f32 color;
    framefunc_get_box2i(&window, &self->window, frame_index);
    framefunc_get_rgba_f32(&color, &self->color, frame_index);
    if (f->format == CVS_FORMAT_F16) {
        rgba_frame_f16 t = { f->data, f->full_window, f->full_window };
        cvs_fill_solid_f16_dev(&t, &window, &color, f->stream);
        f->current_window = t.current_window;
    } else {
        rgba_frame_f32 t = { f->data, f->full_window, f->full_window };
        cvs_fill_solid_f32_dev(&t, &window, &color, f->stream);
        f->current_window = t.current_window;
    }
}
static void solid_render16(PyObject *self, int i, rgba_frame_dev *f) { solid_slot_dev((py_solid *)self, i, f); }
static void solid_slot_16(PyObject *self, int i, rgba_frame_f16 *f) { node_get_frame_host16(self, i, f, CVS_FORMAT_F16, solid_render16); }
static void solid_slot_32(PyObject *self, int i, rgba_frame_f32 *f) { node_get_frame_host32(self, i, f, CVS_FORMAT_F32, solid_render16); }
static video_frame_source_funcs solid_funcs = {
    .flags = VIDEO_SOURCE_FLAG_DEVICE, .get_frame = (video_get_frame_func)solid_slot_16,
    .get_frame_32 = (video_get_frame_32_func)solid_slot_32, .get_frame_dev = (video_get_frame_dev_func)solid_slot_dev };
static PyObject *solid_capsule;
static PyGetSetDef solid_getset[] = { { VIDEO_FRAME_SOURCE_FUNCS, pyext_capsule_getter, NULL, "Video frame source C API.", &solid_capsule }, { NULL } };
static PyTypeObject py_type_Solid = {
    PyVarObject_HEAD_INIT(NULL, 0)
    .tp_name = "fluggo.media.process.SolidColorVideoSource", .tp_basicsize = sizeof(py_solid), .tp_flags = Py_TPFLAGS_DEFAULT,
    .tp_base = &py_type_VideoSource, .tp_new = PyType_GenericNew, .tp_init = (initproc)solid_init,
    .tp_dealloc = (destructor)solid_dealloc, .tp_getset = solid_getset,
};

/* ---------------------------------------------------------------- EmptyVideoSource */

static void empty_slot_dev(PyObject *self, int i, rgba_frame_dev *f) { box2i_set_empty(&f->current_window); }
static void empty_slot_16(PyObject *self, int i, rgba_frame_f16 *f) { box2i_set_empty(&f->current_window); }
static void empty_slot_32(PyObject *self, int i, rgba_frame_f32 *f) { box2i_set_empty(&f->current_window); }
static video_frame_source_funcs empty_funcs = {
    .flags = VIDEO_SOURCE_FLAG_DEVICE, .get_frame = (video_get_frame_func)empty_slot_16,
    .get_frame_32 = (video_get_frame_32_func)empty_slot_32, .get_frame_dev = (video_get_frame_dev_func)empty_slot_dev };
static PyObject *empty_capsule;
static PyGetSetDef empty_getset[] = { { VIDEO_FRAME_SOURCE_FUNCS, pyext_capsule_getter, NULL, "Video frame source C API.", &empty_capsule }, { NULL } };
static PyTypeObject py_type_Empty = {
    PyVarObject_HEAD_INIT(NULL, 0)
    .tp_name = "fluggo.media.process.EmptyVideoSource", .tp_basicsize = sizeof(PyObject), .tp_flags = Py_TPFLAGS_DEFAULT,
    .tp_base = &py_type_VideoSource, .tp_new = PyType_GenericNew, .tp_getset = empty_getset,
};

/* ---------------------------------------------------------------- shared: a node with one upstream source */

typedef struct { PyObject_HEAD pthread_rwlock_t lock; video_source *source; } node1;

static PyObject *node1_get_source(node1 *self, void *c) {
    PyObject *o = self->source ? (PyObject *)self->source->obj : Py_None;
    Py_INCREF(o);
    return o;
}
static PyObject *node1_set_source(node1 *self, PyObject *args) {
    PyObject *src;
    if (!PyArg_ParseTuple(args, "O", &src)) return NULL;
    py_wrlock_nogil(&self->lock);
    bool ok = py_video_take_source(src, &self->source);
    pthread_rwlock_unlock(&self->lock);
    if (!ok) return NULL;
    Py_RETURN_NONE;
}
static int node1_set_source_attr(node1 *self, PyObject *value, void *c) {
    py_wrlock_nogil(&self->lock);
    bool ok = py_video_take_source(value, &self->source);
    pthread_rwlock_unlock(&self->lock);
    return ok ? 0 : -1;
}

/* ---------------------------------------------------------------- VideoGainOffsetFilter */

typedef struct { node1 n; FrameFunctionHolder gain, offset; } py_gain;

static int gain_init(py_gain *self, PyObject *args, PyObject *kw) {
    static char *kwlist[] = { "source", "gain", "offset", NULL };
    PyObject *src, *gain_obj = NULL, *offset_obj = NULL;
    if (!PyArg_ParseTupleAndKeywords(args, kw, "O|OO", kwlist, &src, &gain_obj, &offset_obj)) return -1;
    pthread_rwlock_init(&self->n.lock, NULL);
    framefunc_init(&self->gain, 1.0, 0, 0, 0);
    framefunc_init(&self->offset, 0.0, 0, 0, 0);
    if (!py_video_take_source(src, &self->n.source)) return -1;
    if (gain_obj && !py_framefunc_take_source(gain_obj, &self->gain)) return -1;
    if (offset_obj && !py_framefunc_take_source(offset_obj, &self->offset)) return -1;
    return 0;
}
static void gain_dealloc(py_gain *self) {
    py_video_take_source(NULL, &self->n.source);
    py_framefunc_take_source(NULL, &self->gain);
    py_framefunc_take_source(NULL, &self->offset);
    pthread_rwlock_destroy(&self->n.lock);
    Py_TYPE(self)->tp_free((PyObject *)self);
}
static void gain_render(PyObject *o, int frame_index, rgba_frame_dev *f) {       /* native: f16 */
    py_gain *self = (py_gain *)o;
    rgba_frame_dev in = scratch_like(f, CVS_FORMAT_F16, &f->full_window);
    if (!in.data) { box2i_set_empty(&f->current_window); return; }
    py_rdlock(&self->n.lock);
    pull_dev(self->n.source, frame_index, &in);
    float gain = framefunc_get_f32(&self->gain, frame_index), offset = framefunc_get_f32(&self->offset, frame_index);
    pthread_rwlock_unlock(&self->n.lock);
    rgba_frame_f16 fi = { in.data, in.full_window, in.current_window }, fo = { f->data, f->full_window, f->full_window };
    if (cvs_gain_offset_f16_dev(&fo, &fi, gain, offset, f->stream) != 0) box2i_set_empty(&fo.current_window);
    f->current_window = fo.current_window;
    cvs_pool_free(in.data, f->stream);
}
DEFINE_NODE_VTABLE(gain, CVS_FORMAT_F16, 1, 0)
static void *gain_unused[] UNUSED = { (void *)gain_slot_32 };
static PyObject *holder_get(FrameFunctionHolder *h) {
    if (h->source) { Py_INCREF(h->source); return h->source; }
    return PyFloat_FromDouble(h->constant[0]);
}
static PyObject *gain_get_gain(py_gain *self, void *c) { return holder_get(&self->gain); }
static PyObject *gain_get_offset(py_gain *self, void *c) { return holder_get(&self->offset); }
static int gain_set_gain(py_gain *self, PyObject *v, void *c) { return py_framefunc_take_source(v, &self->gain) ? 0 : -1; }
static int gain_set_offset(py_gain *self, PyObject *v, void *c) { return py_framefunc_take_source(v, &self->offset) ? 0 : -1; }
static PyGetSetDef gain_getset[] = {
    { VIDEO_FRAME_SOURCE_FUNCS, pyext_capsule_getter, NULL, "Video frame source C API.", &gain_capsule },
    { "source", (getter)node1_get_source, (setter)node1_set_source_attr, "The upstream video source." },
    { "gain", (getter)gain_get_gain, (setter)gain_set_gain, "Gain (number or frame function)." },
    { "offset", (getter)gain_get_offset, (setter)gain_set_offset, "Offset (number or frame function)." },
    { NULL }
};
static PyMethodDef node1_methods[] = {
    { "set_source", (PyCFunction)node1_set_source, METH_VARARGS, "set_source(source)" },
    { NULL }
};
static PyTypeObject py_type_Gain = {
    PyVarObject_HEAD_INIT(NULL, 0)
    .tp_name = "fluggo.media.process.VideoGainOffsetFilter", .tp_basicsize = sizeof(py_gain), .tp_flags = Py_TPFLAGS_DEFAULT,
    .tp_base = &py_type_VideoSource, .tp_new = PyType_GenericNew, .tp_init = (initproc)gain_init,
    .tp_dealloc = (destructor)gain_dealloc, .tp_getset = gain_getset, .tp_methods = node1_methods,
};

/* ---------------------------------------------------------------- VideoMixFilter (crossfade) */

typedef struct { PyObject_HEAD video_source *a, *b; FrameFunctionHolder mix_b; } py_mix;

static int mix_init(py_mix *self, PyObject *args, PyObject *kw) {
    static char *kwlist[] = { "src_a", "src_b", "mix_b", NULL };
    PyObject *a, *b, *m;
    if (!PyArg_ParseTupleAndKeywords(args, kw, "OOO", kwlist, &a, &b, &m)) return -1;
    if (!py_video_take_source(a, &self->a) || !py_video_take_source(b, &self->b)) return -1;
    return py_framefunc_take_source(m, &self->mix_b) ? 0 : -1;
}
static void mix_dealloc(py_mix *self) {
    py_video_take_source(NULL, &self->a);
    py_video_take_source(NULL, &self->b);
    py_framefunc_take_source(NULL, &self->mix_b);
    Py_TYPE(self)->tp_free((PyObject *)self);
}
static void mix_render(PyObject *o, int frame_index, rgba_frame_dev *f) {        /* native: f32 */
    py_mix *self = (py_mix *)o;
    /* video_mix_cross_f32_pull (video_mix.c:46-71) on device frames */
    float mix_b = clampf(framefunc_get_f32(&self->mix_b, frame_index), 0.0f, 1.0f);
    if (mix_b == 0.0f) { pull_dev(self->a, frame_index, f); return; }
    if (mix_b == 1.0f) { pull_dev(self->b, frame_index, f); return; }
    rgba_frame_dev tb = scratch_like(f, CVS_FORMAT_F32, &f->full_window);
    if (!tb.data) { box2i_set_empty(&f->current_window); return; }
    pull_dev(self->a, frame_index, f);
    pull_dev(self->b, frame_index, &tb);
    rgba_frame_f32 fa = { f->data, f->full_window, f->current_window }, fb = { tb.data, tb.full_window, tb.current_window };
    if (cvs_mix_cross_f32_dev(&fa, &fa, &fb, mix_b, f->stream) != 0) box2i_set_empty(&fa.current_window);
    f->current_window = fa.current_window;
    cvs_pool_free(tb.data, f->stream);
}
/* a source that fills the f16 host slot only is half-native: its f32 pull is "pull f16, widen" (main.c:105-144) */
static bool half_native(const video_source *src) { return src && src->funcs && src->funcs->get_frame && !src->funcs->get_frame_32; }

/* f16 wanted and both inputs half-native: pull them as f16 and crossfade in one launch (widen, cross, truncate in
 * registers) instead of rendering f32 and narrowing; same arithmetic, a fifth of the traffic */
static bool mix_render_half(py_mix *self, int frame_index, rgba_frame_dev *f) {
    if (f->format != CVS_FORMAT_F16 || !half_native(self->a) || !half_native(self->b)) return false;
    const float mix_b = clampf(framefunc_get_f32(&self->mix_b, frame_index), 0.0f, 1.0f);
    if (mix_b == 0.0f) { pull_dev(self->a, frame_index, f); return true; }       /* video_mix.c:46-71 */
    if (mix_b == 1.0f) { pull_dev(self->b, frame_index, f); return true; }
    rgba_frame_dev ta = scratch_like(f, CVS_FORMAT_F16, &f->full_window), tb = scratch_like(f, CVS_FORMAT_F16, &f->full_window);
    if (ta.data && tb.data) {
        pull_dev(self->a, frame_index, &ta);
        pull_dev(self->b, frame_index, &tb);
        rgba_frame_f16 fa = { ta.data, ta.full_window, ta.current_window }, fb = { tb.data, tb.full_window, tb.current_window };
        rgba_frame_f16 fo = { f->data, f->full_window, f->full_window };
        if (cvs_mix_cross_f16_dev(&fo, &fa, &fb, mix_b, f->stream) != 0) box2i_set_empty(&fo.current_window);
        f->current_window = fo.current_window;
    } else box2i_set_empty(&f->current_window);
    cvs_pool_free(ta.data, f->stream); cvs_pool_free(tb.data, f->stream);
    return true;
}
static void mix_slot_dev(PyObject *self, int i, rgba_frame_dev *f) {
    if (!mix_render_half((py_mix *)self, i, f)) node_get_frame_dev(self, i, f, CVS_FORMAT_F32, mix_render);
}
static void mix_render_any(PyObject *self, int i, rgba_frame_dev *f) { mix_slot_dev(self, i, f); }
static void mix_slot_32(PyObject *self, int i, rgba_frame_f32 *f) { node_get_frame_host32(self, i, f, CVS_FORMAT_F32, mix_render); }
static video_frame_source_funcs mix_funcs = {
    .flags = VIDEO_SOURCE_FLAG_DEVICE, .get_frame_32 = (video_get_frame_32_func)mix_slot_32,
    .get_frame_dev = (video_get_frame_dev_func)mix_slot_dev };
static PyObject *mix_capsule;
static void *mix_unused[] UNUSED = { (void *)mix_render_any };
static PyGetSetDef mix_getset[] = { { VIDEO_FRAME_SOURCE_FUNCS, pyext_capsule_getter, NULL, "Video frame source C API.", &mix_capsule }, { NULL } };
static PyTypeObject py_type_Mix = {
    PyVarObject_HEAD_INIT(NULL, 0)
    .tp_name = "fluggo.media.process.VideoMixFilter", .tp_basicsize = sizeof(py_mix), .tp_flags = Py_TPFLAGS_DEFAULT,
    .tp_base = &py_type_VideoSource, .tp_new = PyType_GenericNew, .tp_init = (initproc)mix_init,
    .tp_dealloc = (destructor)mix_dealloc, .tp_getset = mix_getset,
};

/* ---------------------------------------------------------------- VideoScaler */

typedef struct { node1 n; FrameFunctionHolder target_point, source_point, scale_factors, source_rect; } py_scaler;

static int scaler_init(py_scaler *self, PyObject *args, PyObject *kw) {
    static char *kwlist[] = { "source", "target_point", "source_point", "scale_factors", "source_rect", NULL };
    PyObject *src, *tp, *sp, *sf, *sr;
    if (!PyArg_ParseTupleAndKeywords(args, kw, "OOOOO", kwlist, &src, &tp, &sp, &sf, &sr)) return -1;
    pthread_rwlock_init(&self->n.lock, NULL);
    if (!py_video_take_source(src, &self->n.source)) return -1;
    return (py_framefunc_take_source(tp, &self->target_point) && py_framefunc_take_source(sp, &self->source_point) &&
            py_framefunc_take_source(sf, &self->scale_factors) && py_framefunc_take_source(sr, &self->source_rect)) ? 0 : -1;
}
static void scaler_dealloc(py_scaler *self) {
    py_video_take_source(NULL, &self->n.source);
    py_framefunc_take_source(NULL, &self->target_point); py_framefunc_take_source(NULL, &self->source_point);
    py_framefunc_take_source(NULL, &self->scale_factors); py_framefunc_take_source(NULL, &self->source_rect);
    pthread_rwlock_destroy(&self->n.lock);
    Py_TYPE(self)->tp_free((PyObject *)self);
}
/* fmt: the format of BOTH the pulled input and `f`.  f32 is the node's native format; f16 is taken when the consumer
 * wants f16 and the input is half-native (widen on load, truncate on store inside the scaler's own passes) */
static void scaler_render_fmt(PyObject *o, int frame_index, rgba_frame_dev *f, int fmt) {
    py_scaler *self = (py_scaler *)o;
    py_rdlock(&self->n.lock);                 /* released on every path (the reference leaks it at VideoScaler.c:64-69) */
    if (!self->n.source) { pthread_rwlock_unlock(&self->n.lock); box2i_set_empty(&f->current_window); return; }
    v2f sp, tp, fac; box2i rect;
    framefunc_get_v2f(&sp, &self->source_point, frame_index);
    framefunc_get_v2f(&tp, &self->target_point, frame_index);
    framefunc_get_v2f(&fac, &self->scale_factors, frame_index);
    framefunc_get_box2i(&rect, &self->source_rect, frame_index);
    /* video_scale_bilinear_f32_pull (video_scale.c:288-319) on device frames */
    if (fac.x == 0.0f || fac.y == 0.0f) { box2i_set_empty(&f->current_window); }
    else if (fac.x == 1.0f && fac.y == 1.0f && tp.x == sp.x && tp.y == sp.y) { pull_dev(self->n.source, frame_index, f); }
    else {
        const box2i *tf = &f->full_window;
        box2i need;
        box2i_set(&need, (int)(sp.x - (tp.x - tf->min.x) / fac.x) - 1, (int)(sp.y - (tp.y - tf->min.y) / fac.y) - 1,
                  (int)(sp.x + (tf->max.x - tp.x) / fac.x) + 1, (int)(sp.y + (tf->max.y - tp.y) / fac.y) + 1);
        box2i_intersect(&need, &need, &rect);
        if (box2i_is_empty(&need)) { box2i_set_empty(&f->current_window); }
        else {
            rgba_frame_dev in = scratch_like(f, fmt, &need);
            if (!in.data) box2i_set_empty(&f->current_window);
            else {
                pull_dev(self->n.source, frame_index, &in);
                if (fmt == CVS_FORMAT_F32) {
                    rgba_frame_f32 fs = { in.data, in.full_window, in.current_window }, ft = { f->data, f->full_window, f->full_window };
                    if (cvs_scale_bilinear_f32_dev(&ft, tp, &fs, sp, fac, f->stream) != 0) box2i_set_empty(&ft.current_window);
                    f->current_window = ft.current_window;
                } else {
                    rgba_frame_f16 fs = { in.data, in.full_window, in.current_window }, ft = { f->data, f->full_window, f->full_window };
                    if (cvs_scale_bilinear_f16_dev(&ft, tp, &fs, sp, fac, f->stream) != 0) box2i_set_empty(&ft.current_window);
                    f->current_window = ft.current_window;
                }
                cvs_pool_free(in.data, f->stream);
            }
        }
    }
    pthread_rwlock_unlock(&self->n.lock);
}
static void scaler_render(PyObject *o, int frame_index, rgba_frame_dev *f) { scaler_render_fmt(o, frame_index, f, CVS_FORMAT_F32); }     /* native: f32 */
static void scaler_slot_dev(PyObject *self, int i, rgba_frame_dev *f) {
    py_scaler *sc = (py_scaler *)self;
    py_rdlock(&sc->n.lock);
    const bool half = f->format == CVS_FORMAT_F16 && half_native(sc->n.source);
    pthread_rwlock_unlock(&sc->n.lock);
    if (half) scaler_render_fmt(self, i, f, CVS_FORMAT_F16);
    else node_get_frame_dev(self, i, f, CVS_FORMAT_F32, scaler_render);
}
static void scaler_slot_32(PyObject *self, int i, rgba_frame_f32 *f) { node_get_frame_host32(self, i, f, CVS_FORMAT_F32, scaler_render); }
static video_frame_source_funcs scaler_funcs = {
    .flags = VIDEO_SOURCE_FLAG_DEVICE, .get_frame_32 = (video_get_frame_32_func)scaler_slot_32,
    .get_frame_dev = (video_get_frame_dev_func)scaler_slot_dev };
static PyObject *scaler_capsule;
static PyObject *scaler_source(py_scaler *self, PyObject *dummy) { return node1_get_source(&self->n, NULL); }
static PyMethodDef scaler_methods[] = {
    { "source", (PyCFunction)scaler_source, METH_NOARGS, "Gets the video source." },
    { "set_source", (PyCFunction)node1_set_source, METH_VARARGS, "Sets the video source." },
    { NULL }
};
static PyGetSetDef scaler_getset[] = { { VIDEO_FRAME_SOURCE_FUNCS, pyext_capsule_getter, NULL, "Video frame source C API.", &scaler_capsule }, { NULL } };
static PyTypeObject py_type_Scaler = {
    PyVarObject_HEAD_INIT(NULL, 0)
    .tp_name = "fluggo.media.process.VideoScaler", .tp_basicsize = sizeof(py_scaler), .tp_flags = Py_TPFLAGS_DEFAULT,
    .tp_base = &py_type_VideoSource, .tp_new = PyType_GenericNew, .tp_init = (initproc)scaler_init,
    .tp_dealloc = (destructor)scaler_dealloc, .tp_getset = scaler_getset, .tp_methods = scaler_methods,
};

/* ---------------------------------------------------------------- VideoPassThroughFilter (subclassable) */

typedef struct { node1 n; int offset, start_frame, end_frame; bool has_start, has_end; } py_pass;

static int pass_init(py_pass *self, PyObject *args, PyObject *kw) {
    static char *kwlist[] = { "source", "offset", "start_frame", "end_frame", NULL };
    PyObject *src = Py_None, *start = Py_None, *end = Py_None;
    self->offset = 0;
    if (!PyArg_ParseTupleAndKeywords(args, kw, "|OiOO", kwlist, &src, &self->offset, &start, &end)) return -1;
    pthread_rwlock_init(&self->n.lock, NULL);
    self->has_start = start != Py_None;
    self->has_end = end != Py_None;
    if (self->has_start) { self->start_frame = (int)PyLong_AsLong(start); if (PyErr_Occurred()) return -1; }
    if (self->has_end) { self->end_frame = (int)PyLong_AsLong(end); if (PyErr_Occurred()) return -1; }
    return py_video_take_source(src, &self->n.source) ? 0 : -1;
}
static void pass_dealloc(py_pass *self) {
    py_video_take_source(NULL, &self->n.source);
    pthread_rwlock_destroy(&self->n.lock);
    Py_TYPE(self)->tp_free((PyObject *)self);
}
/* forwards in the caller's format: no native format of its own (VideoPassThroughFilter.c:70-119) */
static void pass_slot_dev(py_pass *self, int frame_index, rgba_frame_dev *f) {
    py_rdlock(&self->n.lock);
    if ((self->has_start && frame_index < self->start_frame) || (self->has_end && frame_index >= self->end_frame)) box2i_set_empty(&f->current_window);
    else pull_dev(self->n.source, frame_index + self->offset, f);
    pthread_rwlock_unlock(&self->n.lock);
}
static void pass_render(PyObject *self, int i, rgba_frame_dev *f) { pass_slot_dev((py_pass *)self, i, f); }
static void pass_slot_16(PyObject *self, int i, rgba_frame_f16 *f) { node_get_frame_host16(self, i, f, CVS_FORMAT_F16, pass_render); }
static void pass_slot_32(PyObject *self, int i, rgba_frame_f32 *f) { node_get_frame_host32(self, i, f, CVS_FORMAT_F32, pass_render); }
static video_frame_source_funcs pass_funcs = {
    .flags = VIDEO_SOURCE_FLAG_DEVICE, .get_frame = (video_get_frame_func)pass_slot_16,
    .get_frame_32 = (video_get_frame_32_func)pass_slot_32, .get_frame_dev = (video_get_frame_dev_func)pass_slot_dev };
static PyObject *pass_capsule;
static PyObject *pass_get_offset(py_pass *self, void *c) { return PyLong_FromLong(self->offset); }
static int pass_set_offset(py_pass *self, PyObject *v, void *c) {
    long x = PyLong_AsLong(v);
    if (PyErr_Occurred()) return -1;
    py_wrlock_nogil(&self->n.lock); self->offset = (int)x; pthread_rwlock_unlock(&self->n.lock);
    return 0;
}
static PyObject *opt_int(bool has, int v) { if (!has) Py_RETURN_NONE; return PyLong_FromLong(v); }
static PyObject *pass_get_start(py_pass *self, void *c) { return opt_int(self->has_start, self->start_frame); }
static PyObject *pass_get_end(py_pass *self, void *c) { return opt_int(self->has_end, self->end_frame); }
static int pass_set_bound(py_pass *self, PyObject *v, bool *has, int *slot) {
    long x = 0;
    if (v && v != Py_None) { x = PyLong_AsLong(v); if (PyErr_Occurred()) return -1; }
    py_wrlock_nogil(&self->n.lock);
    *has = v && v != Py_None; *slot = (int)x;
    pthread_rwlock_unlock(&self->n.lock);
    return 0;
}
static int pass_set_start(py_pass *self, PyObject *v, void *c) { return pass_set_bound(self, v, &self->has_start, &self->start_frame); }
static int pass_set_end(py_pass *self, PyObject *v, void *c) { return pass_set_bound(self, v, &self->has_end, &self->end_frame); }
static PyGetSetDef pass_getset[] = {
    { VIDEO_FRAME_SOURCE_FUNCS, pyext_capsule_getter, NULL, "Video frame source C API.", &pass_capsule },
    { "source", (getter)node1_get_source, (setter)node1_set_source_attr, "The upstream video source." },
    { "offset", (getter)pass_get_offset, (setter)pass_set_offset, "Offset added to the frame index." },
    { "start_frame", (getter)pass_get_start, (setter)pass_set_start, "First frame passed through, or None." },
    { "end_frame", (getter)pass_get_end, (setter)pass_set_end, "First frame no longer passed through, or None." },
    { NULL }
};
static PyTypeObject py_type_Pass = {
    PyVarObject_HEAD_INIT(NULL, 0)
    .tp_name = "fluggo.media.process.VideoPassThroughFilter", .tp_basicsize = sizeof(py_pass),
    .tp_flags = Py_TPFLAGS_DEFAULT | Py_TPFLAGS_BASETYPE,           /* plugins.VideoStream subclasses it (_source.py:399) */
    .tp_base = &py_type_VideoSource, .tp_new = PyType_GenericNew, .tp_init = (initproc)pass_init,
    .tp_dealloc = (destructor)pass_dealloc, .tp_getset = pass_getset, .tp_methods = node1_methods,
};

/* ---------------------------------------------------------------- VideoSequence */

typedef struct { PyObject *tuple; video_source *source; int length, offset, start_frame; } seq_elem;
typedef struct { PyObject_HEAD pthread_rwlock_t lock; seq_elem *e; Py_ssize_t count, cap; } py_seq;

static int seq_init(py_seq *self, PyObject *args, PyObject *kw) { pthread_rwlock_init(&self->lock, NULL); return 0; }
static void seq_dealloc(py_seq *self) {
    for (Py_ssize_t i = 0; i < self->count; i++) { Py_CLEAR(self->e[i].tuple); py_video_take_source(NULL, &self->e[i].source); }
    PyMem_Free(self->e);
    pthread_rwlock_destroy(&self->lock);
    Py_TYPE(self)->tp_free((PyObject *)self);
}
static void seq_restart(py_seq *self) {
    int at = 0;
    for (Py_ssize_t i = 0; i < self->count; i++) { self->e[i].start_frame = at; at += self->e[i].length; }
}
/* the element covering frame_index, or NULL (VideoSequence.c:58-82; negative frames and gaps are empty) */
static seq_elem *seq_pick(py_seq *self, int frame_index) {
    if (frame_index < 0) return NULL;
    Py_ssize_t lo = 0, hi = self->count;
    while (lo < hi) {
        Py_ssize_t mid = (lo + hi) / 2;
        if (frame_index >= self->e[mid].start_frame + self->e[mid].length) lo = mid + 1; else hi = mid;
    }
    if (lo >= self->count || !self->e[lo].source || frame_index < self->e[lo].start_frame) return NULL;
    return &self->e[lo];
}
static void seq_slot_dev(py_seq *self, int frame_index, rgba_frame_dev *f) {
    py_rdlock(&self->lock);
    seq_elem *el = seq_pick(self, frame_index);
    if (!el) box2i_set_empty(&f->current_window);
    else pull_dev(el->source, frame_index - el->start_frame + el->offset, f);
    pthread_rwlock_unlock(&self->lock);
}
static void seq_render(PyObject *self, int i, rgba_frame_dev *f) { seq_slot_dev((py_seq *)self, i, f); }
static void seq_slot_16(PyObject *self, int i, rgba_frame_f16 *f) { node_get_frame_host16(self, i, f, CVS_FORMAT_F16, seq_render); }
static void seq_slot_32(PyObject *self, int i, rgba_frame_f32 *f) { node_get_frame_host32(self, i, f, CVS_FORMAT_F32, seq_render); }
static video_frame_source_funcs seq_funcs = {
    .flags = VIDEO_SOURCE_FLAG_DEVICE, .get_frame = (video_get_frame_func)seq_slot_16,
    .get_frame_32 = (video_get_frame_32_func)seq_slot_32, .get_frame_dev = (video_get_frame_dev_func)seq_slot_dev };
static PyObject *seq_capsule;

static Py_ssize_t seq_len(py_seq *self) { return self->count; }
static PyObject *seq_item(py_seq *self, Py_ssize_t i) {
    if (i < 0 || i >= self->count) { PyErr_SetString(PyExc_IndexError, "Index was out of range."); return NULL; }
    Py_INCREF(self->e[i].tuple);
    return self->e[i].tuple;
}
static bool seq_parse(PyObject *v, seq_elem *out) {
    PyObject *src; int offset, length;
    memset(out, 0, sizeof *out);
    if (!PyArg_ParseTuple(v, "Oii", &src, &offset, &length)) return false;
    if (length < 0) { PyErr_SetString(PyExc_ValueError, "Length cannot be less than zero."); return false; }
    if (!py_video_take_source(src, &out->source)) return false;
    Py_INCREF(v);
    out->tuple = v; out->offset = offset; out->length = length;
    return true;
}
static int seq_ass_item(py_seq *self, Py_ssize_t i, PyObject *v) {
    if (i < 0 || i >= self->count) { PyErr_SetString(PyExc_IndexError, "Index was out of range."); return -1; }
    seq_elem fresh;
    if (v && !seq_parse(v, &fresh)) return -1;
    py_wrlock_nogil(&self->lock);
    Py_CLEAR(self->e[i].tuple);
    py_video_take_source(NULL, &self->e[i].source);
    if (v) self->e[i] = fresh;
    else { memmove(&self->e[i], &self->e[i + 1], sizeof(seq_elem) * (size_t)(self->count - i - 1)); self->count--; }
    seq_restart(self);
    pthread_rwlock_unlock(&self->lock);
    return 0;
}
static PyObject *seq_insert_at(py_seq *self, Py_ssize_t i, PyObject *v) {
    if (i < 0) i += self->count;
    if (i < 0) i = 0;
    if (i > self->count) { PyErr_SetString(PyExc_IndexError, "Index was out of range."); return NULL; }
    seq_elem fresh;
    if (!seq_parse(v, &fresh)) return NULL;
    py_wrlock_nogil(&self->lock);
    if (self->count == self->cap) {
        Py_ssize_t cap = self->cap ? self->cap * 2 : 8;
        seq_elem *e = PyMem_Realloc(self->e, sizeof(seq_elem) * (size_t)cap);
        if (!e) { pthread_rwlock_unlock(&self->lock); Py_DECREF(fresh.tuple); py_video_take_source(NULL, &fresh.source); return PyErr_NoMemory(); }
        self->e = e; self->cap = cap;
    }
    memmove(&self->e[i + 1], &self->e[i], sizeof(seq_elem) * (size_t)(self->count - i));
    self->e[i] = fresh;
    self->count++;
    seq_restart(self);
    pthread_rwlock_unlock(&self->lock);
    Py_RETURN_NONE;
}
static PyObject *seq_insert(py_seq *self, PyObject *args) {
    Py_ssize_t i; PyObject *v;
    if (!PyArg_ParseTuple(args, "nO", &i, &v)) return NULL;
    return seq_insert_at(self, i, v);
}
static PyObject *seq_append(py_seq *self, PyObject *args) {
    PyObject *v;
    if (!PyArg_ParseTuple(args, "O", &v)) return NULL;
    return seq_insert_at(self, self->count, v);
}
static PyObject *seq_start_frame(py_seq *self, PyObject *args) {
    Py_ssize_t i;
    if (!PyArg_ParseTuple(args, "n", &i)) return NULL;
    if (i < 0 || i >= self->count) { PyErr_SetString(PyExc_IndexError, "Index was out of range."); return NULL; }
    return PyLong_FromLong(self->e[i].start_frame);
}
static PySequenceMethods seq_as_sequence = { .sq_length = (lenfunc)seq_len, .sq_item = (ssizeargfunc)seq_item, .sq_ass_item = (ssizeobjargproc)seq_ass_item };
static PyMethodDef seq_methods[] = {
    { "insert", (PyCFunction)seq_insert, METH_VARARGS, "insert(index, (source, offset, length))" },
    { "append", (PyCFunction)seq_append, METH_VARARGS, "append((source, offset, length))" },
    { "get_start_frame", (PyCFunction)seq_start_frame, METH_VARARGS, "get_start_frame(index) -> first frame of that element" },
    { NULL }
};
static PyGetSetDef seq_getset[] = { { VIDEO_FRAME_SOURCE_FUNCS, pyext_capsule_getter, NULL, "Video frame source C API.", &seq_capsule }, { NULL } };
static PyTypeObject py_type_Seq = {
    PyVarObject_HEAD_INIT(NULL, 0)
    .tp_name = "fluggo.media.process.VideoSequence", .tp_basicsize = sizeof(py_seq), .tp_flags = Py_TPFLAGS_DEFAULT,
    .tp_base = &py_type_VideoSource, .tp_new = PyType_GenericNew, .tp_init = (initproc)seq_init, .tp_dealloc = (destructor)seq_dealloc,
    .tp_getset = seq_getset, .tp_methods = seq_methods, .tp_as_sequence = &seq_as_sequence,
};

/* ---------------------------------------------------------------- Pulldown23RemovalFilter(source, offset)
 * src/process/Pulldown23RemovalFilter.c:31-107: 2:3 pulldown removal (30 -> 24 frames/s).  Four of every five source
 * frames map to output frames whole; the fifth output frame is woven from the odd rows of one source frame and the even
 * rows of the next.  Index arithmetic and weave are the library's (cvs_pulldown23_frames, cvs_weave_fields_f16_dev);
 * both source frames are pulled into device frames, nothing crosses PCIe. */

typedef struct { node1 n; int offset; } py_pulldown;

static int pulldown_init(py_pulldown *self, PyObject *args, PyObject *kw) {
    PyObject *src;
    if (!PyArg_ParseTuple(args, "Oi", &src, &self->offset)) return -1;
    pthread_rwlock_init(&self->n.lock, NULL);
    return py_video_take_source(src, &self->n.source) ? 0 : -1;
}
static void pulldown_dealloc(py_pulldown *self) {
    py_video_take_source(NULL, &self->n.source);
    pthread_rwlock_destroy(&self->n.lock);
    Py_TYPE(self)->tp_free((PyObject *)self);
}
static void pulldown_render(PyObject *o, int frame_index, rgba_frame_dev *f) {      /* native: f16 */
    py_pulldown *self = (py_pulldown *)o;
    py_rdlock(&self->n.lock);
    if (!self->n.source) { box2i_set_empty(&f->current_window); pthread_rwlock_unlock(&self->n.lock); return; }
    int first, second;
    const int mixed = cvs_pulldown23_frames(self->offset, frame_index, &first, &second);
    pull_dev(self->n.source, first, f);
    if (mixed && !box2i_is_empty(&f->current_window)) {
        rgba_frame_dev other = scratch_like(f, CVS_FORMAT_F16, &f->current_window);       /* :92-95: a buffer for exactly that window */
        if (other.data) {
            pull_dev(self->n.source, second, &other);
            rgba_frame_f16 frame = { f->data, f->full_window, f->current_window }, field = { other.data, other.full_window, other.current_window };
            if (cvs_weave_fields_f16_dev(&frame, &field, f->stream) != 0) box2i_set_empty(&f->current_window);
            cvs_pool_free(other.data, f->stream);
        } else box2i_set_empty(&f->current_window);
    }
    pthread_rwlock_unlock(&self->n.lock);
}
DEFINE_NODE_VTABLE(pulldown, CVS_FORMAT_F16, 1, 0)
static void *pulldown_unused[] UNUSED = { (void *)pulldown_slot_32 };
static PyGetSetDef pulldown_getset[] = { { VIDEO_FRAME_SOURCE_FUNCS, pyext_capsule_getter, NULL, "Video frame source C API.", &pulldown_capsule }, { NULL } };
static PyTypeObject py_type_Pulldown = {
    PyVarObject_HEAD_INIT(NULL, 0)
    .tp_name = "fluggo.media.process.Pulldown23RemovalFilter", .tp_basicsize = sizeof(py_pulldown), .tp_flags = Py_TPFLAGS_DEFAULT,
    .tp_base = &py_type_VideoSource, .tp_new = PyType_GenericNew, .tp_init = (initproc)pulldown_init,
    .tp_dealloc = (destructor)pulldown_dealloc, .tp_getset = pulldown_getset,
};

int init_sources(PyObject *module) {
    if (pyext_make_capsule(&solid_capsule, &solid_funcs) < 0 || pyext_make_capsule(&empty_capsule, &empty_funcs) < 0 ||
        pyext_make_capsule(&gain_capsule, &gain_funcs) < 0 || pyext_make_capsule(&mix_capsule, &mix_funcs) < 0 ||
        pyext_make_capsule(&scaler_capsule, &scaler_funcs) < 0 || pyext_make_capsule(&pass_capsule, &pass_funcs) < 0 ||
        pyext_make_capsule(&seq_capsule, &seq_funcs) < 0 || pyext_make_capsule(&pulldown_capsule, &pulldown_funcs) < 0) return -1;
    if (pyext_add_type(module, "SolidColorVideoSource", &py_type_Solid) < 0 || pyext_add_type(module, "EmptyVideoSource", &py_type_Empty) < 0 ||
        pyext_add_type(module, "VideoGainOffsetFilter", &py_type_Gain) < 0 || pyext_add_type(module, "VideoMixFilter", &py_type_Mix) < 0 ||
        pyext_add_type(module, "VideoScaler", &py_type_Scaler) < 0 || pyext_add_type(module, "VideoPassThroughFilter", &py_type_Pass) < 0 ||
        pyext_add_type(module, "VideoSequence", &py_type_Seq) < 0 || pyext_add_type(module, "Pulldown23RemovalFilter", &py_type_Pulldown) < 0) return -1;
    return 0;
}
