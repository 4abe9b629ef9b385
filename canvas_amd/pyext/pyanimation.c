/*
 * pyanimation.c -- AnimationFunc / AnimationPoint: a frame function defined by key points.
 *
 * Behaviour follows src/process/AnimationFunc.c:
 *   AnimationPoint(type, frame, value)      type POINT_HOLD (0) or POINT_LINEAR (1), fractional frame,
 *                                           value a number or a 1..4-tuple (missing slots are 0)   (:104-120)
 *   .value -> 4-tuple, .frame (settable: the owner re-sorts), .type                                  (:122-157)
 *   AnimationFunc(): add(type, frame, value) | add(point) -> point, remove(point), len(), [i]       (:283-349)
 *   evaluation (:408-463): no points -> zeros; before the first point -> its value; after the last
 *   point, or the left neighbour is HOLD -> the left value; LINEAR ->
 *       (right * (f - left.frame) + left * (right.frame - f)) / (right.frame - left.frame)      in doubles.
 *   Among points with equal frames the one reached last by a forward scan is the left neighbour (:389-399).
 *
 * The reference keeps a GSequence plus a cursor that remembers the last lookup; here the points live in a
 * sorted array under a pthread rwlock and every lookup is a binary search, so readers never write shared
 * state and any number of render threads can evaluate one function at a time.
 */
#include "pyext.h"
#include <pthread.h>

enum { POINT_HOLD = 0, POINT_LINEAR = 1, POINT_MAX = 1 };

struct py_anim;
typedef struct {
    PyObject_HEAD
    struct py_anim *owner;     /* strong reference while the point is in a function */
    int type;
    double frame;
    double values[4];
} py_anim_point;

typedef struct py_anim {
    PyObject_HEAD
    py_anim_point **points;    /* sorted by frame, stable for equal frames; each entry holds a reference */
    Py_ssize_t count, cap;
    pthread_rwlock_t lock;
    int lock_ready;
} py_anim;

static PyTypeObject py_type_AnimationPoint;

static bool parse_value(double out[4], PyObject *src) {
    out[0] = out[1] = out[2] = out[3] = 0.0;
    if (PyTuple_Check(src)) {
        Py_ssize_t n = PyTuple_GET_SIZE(src);
        if (n == 0) { PyErr_SetString(PyExc_ValueError, "An empty tuple was passed."); return false; }
        if (n > 4) { PyErr_Format(PyExc_ValueError, "One of the tuples passed has more than four entries (%zd).", n); return false; }
        for (Py_ssize_t i = 0; i < n; i++) {
            out[i] = PyFloat_AsDouble(PyTuple_GET_ITEM(src, i));
            if (out[i] == -1.0 && PyErr_Occurred()) return false;
        }
        return true;
    }
    out[0] = PyFloat_AsDouble(src);
    return !(out[0] == -1.0 && PyErr_Occurred());
}

static int point_init(py_anim_point *self, PyObject *args, PyObject *kw) {
    static char *kwlist[] = { "type", "frame", "value", NULL };
    PyObject *value;
    if (!PyArg_ParseTupleAndKeywords(args, kw, "idO", kwlist, &self->type, &self->frame, &value)) return -1;
    if (self->type < 0 || self->type > POINT_MAX) { PyErr_SetString(PyExc_Exception, "The given type value was invalid."); return -1; }
    return parse_value(self->values, value) ? 0 : -1;
}

/* first index whose frame is greater than `frame` (points with the same frame stay in insertion order) */
static Py_ssize_t upper_bound(const py_anim *a, double frame) {
    Py_ssize_t lo = 0, hi = a->count;
    while (lo < hi) {
        Py_ssize_t mid = lo + (hi - lo) / 2;
        if (a->points[mid]->frame <= frame) lo = mid + 1; else hi = mid;
    }
    return lo;
}

static Py_ssize_t index_of(const py_anim *a, const py_anim_point *p) {
    for (Py_ssize_t i = 0; i < a->count; i++) if (a->points[i] == p) return i;
    return -1;
}

static PyObject *point_get_value(py_anim_point *self, void *c) { return Py_BuildValue("dddd", self->values[0], self->values[1], self->values[2], self->values[3]); }
static PyObject *point_get_frame(py_anim_point *self, void *c) { return PyFloat_FromDouble(self->frame); }
static PyObject *point_get_type(py_anim_point *self, void *c) { return PyLong_FromLong(self->type); }

static int point_set_frame(py_anim_point *self, PyObject *value, void *c) {
    if (!value) { PyErr_SetString(PyExc_TypeError, "cannot delete frame"); return -1; }
    double frame = PyFloat_AsDouble(value);
    if (frame == -1.0 && PyErr_Occurred()) return -1;
    py_anim *a = self->owner;
    if (!a) { self->frame = frame; return 0; }
    py_wrlock_nogil(&a->lock);
    Py_ssize_t at = index_of(a, self);
    if (at >= 0) {          /* take it out, find its new place, put it back */
        memmove(a->points + at, a->points + at + 1, sizeof(*a->points) * (size_t)(a->count - at - 1));
        a->count--;
        self->frame = frame;
        Py_ssize_t to = upper_bound(a, frame);
        memmove(a->points + to + 1, a->points + to, sizeof(*a->points) * (size_t)(a->count - to));
        a->points[to] = self;
        a->count++;
    } else self->frame = frame;
    pthread_rwlock_unlock(&a->lock);
    return 0;
}

static PyGetSetDef point_getset[] = {
    { "value", (getter)point_get_value, NULL, "Value at this point in the animation." },
    { "frame", (getter)point_get_frame, (setter)point_set_frame, "Frame for this animation point, which may be fractional." },
    { "type", (getter)point_get_type, NULL, "Type of this point." },
    { NULL }
};

static int point_traverse(py_anim_point *self, visitproc visit, void *arg) { Py_VISIT((PyObject *)self->owner); return 0; }
static int point_clear(py_anim_point *self) { Py_CLEAR(self->owner); return 0; }
static void point_dealloc(py_anim_point *self) {
    PyObject_GC_UnTrack(self);
    Py_CLEAR(self->owner);
    Py_TYPE(self)->tp_free((PyObject *)self);
}

static PyTypeObject py_type_AnimationPoint = {
    PyVarObject_HEAD_INIT(NULL, 0)
    .tp_name = "fluggo.media.process.AnimationPoint", .tp_basicsize = sizeof(py_anim_point),
    .tp_flags = Py_TPFLAGS_DEFAULT | Py_TPFLAGS_HAVE_GC | Py_TPFLAGS_BASETYPE, .tp_new = PyType_GenericNew,
    .tp_dealloc = (destructor)point_dealloc, .tp_init = (initproc)point_init, .tp_traverse = (traverseproc)point_traverse,
    .tp_clear = (inquiry)point_clear, .tp_getset = point_getset,
};

/* ------------------------------------------------------------------ the function */

static int anim_init(py_anim *self, PyObject *args, PyObject *kw) {
    if (!self->lock_ready) { pthread_rwlock_init(&self->lock, NULL); self->lock_ready = 1; }
    return 0;
}

static void anim_drop_all(py_anim *self) {
    for (Py_ssize_t i = 0; i < self->count; i++) {
        py_anim_point *p = self->points[i];
        Py_CLEAR(p->owner);
        Py_DECREF(p);
    }
    self->count = 0;
}

static int anim_traverse(py_anim *self, visitproc visit, void *arg) {
    for (Py_ssize_t i = 0; i < self->count; i++) Py_VISIT((PyObject *)self->points[i]);
    return 0;
}
static int anim_clear(py_anim *self) { anim_drop_all(self); return 0; }
static void anim_dealloc(py_anim *self) {
    PyObject_GC_UnTrack(self);
    anim_drop_all(self);
    PyMem_Free(self->points);
    if (self->lock_ready) pthread_rwlock_destroy(&self->lock);
    Py_TYPE(self)->tp_free((PyObject *)self);
}

static Py_ssize_t anim_len(py_anim *self) { return self->count; }
static PyObject *anim_item(py_anim *self, Py_ssize_t i) {
    if (i < 0 || i >= self->count) { PyErr_SetString(PyExc_IndexError, "Index was out of range."); return NULL; }
    Py_INCREF(self->points[i]);
    return (PyObject *)self->points[i];
}
static PySequenceMethods anim_sequence = { .sq_length = (lenfunc)anim_len, .sq_item = (ssizeargfunc)anim_item };

static PyObject *anim_add(py_anim *self, PyObject *args, PyObject *kw) {
    py_anim_point *p;
    if (!self->lock_ready) { pthread_rwlock_init(&self->lock, NULL); self->lock_ready = 1; }
    if (PyTuple_GET_SIZE(args) == 1 && !kw) {
        if (!PyArg_ParseTuple(args, "O!", &py_type_AnimationPoint, &p)) return NULL;
        if (p->owner) { PyErr_SetString(PyExc_Exception, "This point already belongs to an animation."); return NULL; }
        Py_INCREF(p);
    } else {
        p = (py_anim_point *)PyObject_Call((PyObject *)&py_type_AnimationPoint, args, kw);
        if (!p) return NULL;
    }
    py_wrlock_nogil(&self->lock);
    if (self->count == self->cap) {
        Py_ssize_t cap = self->cap ? self->cap * 2 : 8;
        py_anim_point **grown = PyMem_Realloc(self->points, sizeof(*grown) * (size_t)cap);
        if (!grown) { pthread_rwlock_unlock(&self->lock); Py_DECREF(p); return PyErr_NoMemory(); }
        self->points = grown; self->cap = cap;
    }
    Py_ssize_t to = upper_bound(self, p->frame);
    memmove(self->points + to + 1, self->points + to, sizeof(*self->points) * (size_t)(self->count - to));
    self->points[to] = p;           /* the array keeps the reference made above */
    self->count++;
    pthread_rwlock_unlock(&self->lock);
    Py_INCREF(self);
    p->owner = self;
    Py_INCREF(p);
    return (PyObject *)p;
}

static PyObject *anim_remove(py_anim *self, PyObject *args) {
    py_anim_point *p;
    if (!PyArg_ParseTuple(args, "O!", &py_type_AnimationPoint, &p)) return NULL;
    if (p->owner != self) Py_RETURN_NONE;
    py_wrlock_nogil(&self->lock);
    Py_ssize_t at = index_of(self, p);
    if (at >= 0) {
        memmove(self->points + at, self->points + at + 1, sizeof(*self->points) * (size_t)(self->count - at - 1));
        self->count--;
    }
    pthread_rwlock_unlock(&self->lock);
    Py_CLEAR(p->owner);
    if (at >= 0) Py_DECREF(p);
    Py_RETURN_NONE;
}

static void anim_values(py_anim *self, ssize_t count, double *frames, double (*out)[4]) {
    if (!self->lock_ready) { for (ssize_t i = 0; i < count; i++) out[i][0] = out[i][1] = out[i][2] = out[i][3] = 0.0; return; }
    py_rdlock(&self->lock);
    for (ssize_t i = 0; i < count; i++) {
        const double f = frames[i];
        const Py_ssize_t r = upper_bound(self, f);
        const py_anim_point *left = r > 0 ? self->points[r - 1] : NULL;
        const py_anim_point *right = r < self->count ? self->points[r] : NULL;
        if (!left && !right) { out[i][0] = out[i][1] = out[i][2] = out[i][3] = 0.0; }
        else if (!left) memcpy(out[i], right->values, sizeof right->values);
        else if (!right || left->type == POINT_HOLD) memcpy(out[i], left->values, sizeof left->values);
        else {
            const double distance = right->frame - left->frame;
            for (int k = 0; k < 4; k++)
                out[i][k] = (right->values[k] * (f - left->frame) + left->values[k] * (right->frame - f)) / distance;
        }
    }
    pthread_rwlock_unlock(&self->lock);
}

static PyObject *anim_capsule;
static FrameFunctionFuncs anim_funcs = { 0, (framefunc_get_values_func)anim_values };
static PyGetSetDef anim_getset[] = { { FRAME_FUNCTION_FUNCS, pyext_capsule_getter, NULL, "Frame function C API.", &anim_capsule }, { NULL } };
static PyMethodDef anim_methods[] = {
    { "add", (PyCFunction)anim_add, METH_VARARGS | METH_KEYWORDS, "point = func.add(type, frame, value) or func.add(point): adds a point to the animation." },
    { "remove", (PyCFunction)anim_remove, METH_VARARGS, "func.remove(point): removes a point from the animation." },
    { NULL }
};

CVS_EXPORT PyTypeObject py_type_AnimationFunc = {
    PyVarObject_HEAD_INIT(NULL, 0)
    .tp_name = "fluggo.media.process.AnimationFunc", .tp_basicsize = sizeof(py_anim),
    .tp_base = &py_type_FrameFunction, .tp_flags = Py_TPFLAGS_DEFAULT | Py_TPFLAGS_HAVE_GC, .tp_new = PyType_GenericNew,
    .tp_dealloc = (destructor)anim_dealloc, .tp_init = (initproc)anim_init, .tp_traverse = (traverseproc)anim_traverse,
    .tp_clear = (inquiry)anim_clear, .tp_getset = anim_getset, .tp_methods = anim_methods, .tp_as_sequence = &anim_sequence,
};

int init_animation(PyObject *module) {
    anim_capsule = PyCapsule_New(&anim_funcs, FRAME_FUNCTION_FUNCS, NULL);
    if (!anim_capsule) return -1;
    if (pyext_add_type(module, "AnimationPoint", &py_type_AnimationPoint) < 0) return -1;
    if (pyext_add_type(module, "AnimationFunc", &py_type_AnimationFunc) < 0) return -1;
    if (PyModule_AddIntConstant(module, "POINT_HOLD", POINT_HOLD) < 0) return -1;
    return PyModule_AddIntConstant(module, "POINT_LINEAR", POINT_LINEAR);
}
