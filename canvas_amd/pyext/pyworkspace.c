/*
 * pyworkspace.c -- VideoWorkspace (+ item objects) and VideoPullQueue.
 *
 * VideoWorkspace: src/process/VideoWorkspace.c:205-262,325-330,406-472 --
 *   add(source=, offset=, x=, length=, z=, tag=) -> item;  remove(item);  len(ws);  ws[i];
 *   item.x .length .z .offset .source .tag,  item.update(**kw).
 *   The compositor itself is the library's workspace_t (csrc/host/workspace.c); its vtable already
 *   has the device slot, so the whole over stack runs in HBM.
 * VideoPullQueue: src/process/VideoPullQueue.c:72-197 -- enqueue(source, frame_index, window,
 *   callback, user_data) -> item with cancel(); two worker threads pull frames without the GIL.
 *   The reference hands the callback to the glib main loop; there is no glib here, so the callback runs
 *   on the worker thread that produced the frame, under the GIL (callback(frame_index, frame, user_data)).
 */
#include "pyext.h"

/* ---------------------------------------------------------------- VideoWorkspace */

typedef struct { PyObject_HEAD pthread_rwlock_t lock; workspace_t *ws; video_source source; PyObject *items; } py_workspace;
typedef struct { PyObject_HEAD py_workspace *owner; workspace_item_t *item; video_source *source; } py_ws_item;

static PyTypeObject py_type_WorkspaceItem;

static int ws_init(py_workspace *self, PyObject *args, PyObject *kw) {
    pthread_rwlock_init(&self->lock, NULL);
    self->ws = workspace_create();
    self->items = PyList_New(0);
    if (!self->ws || !self->items) { PyErr_NoMemory(); return -1; }
    workspace_as_video_source(self->ws, &self->source);
    return 0;
}

static void item_detach(py_ws_item *it) {
    if (it->item) { workspace_remove_item(it->item); it->item = NULL; }
    py_video_take_source(NULL, &it->source);
    it->owner = NULL;
}

static void ws_dealloc(py_workspace *self) {
    if (self->items) {
        for (Py_ssize_t i = 0; i < PyList_GET_SIZE(self->items); i++) item_detach((py_ws_item *)PyList_GET_ITEM(self->items, i));
        Py_CLEAR(self->items);
    }
    if (self->ws) workspace_free(self->ws);
    pthread_rwlock_destroy(&self->lock);
    Py_TYPE(self)->tp_free((PyObject *)self);
}

/* slots forward to the library's workspace vtable under the reader lock (VideoWorkspace.c:325-330) */
static void ws_slot_32(py_workspace *self, int frame_index, rgba_frame_f32 *f) {
    py_rdlock(&self->lock);
    video_get_frame_f32(&self->source, frame_index, f);
    pthread_rwlock_unlock(&self->lock);
}
static void ws_slot_dev(py_workspace *self, int frame_index, rgba_frame_dev *f) {
    py_rdlock(&self->lock);
    video_get_frame_dev(&self->source, frame_index, f);
    pthread_rwlock_unlock(&self->lock);
}
static void ws_render(PyObject *self, int i, rgba_frame_dev *f) { ws_slot_dev((py_workspace *)self, i, f); }
static void ws_host_32(PyObject *self, int i, rgba_frame_f32 *f) { node_get_frame_host32(self, i, f, CVS_FORMAT_F32, ws_render); }
static video_frame_source_funcs ws_funcs = {
    .flags = VIDEO_SOURCE_FLAG_DEVICE, .get_frame_32 = (video_get_frame_32_func)ws_host_32,
    .get_frame_dev = (video_get_frame_dev_func)ws_slot_dev };
static void *ws_unused[] __attribute__((unused)) = { (void *)ws_slot_32 };
static PyObject *ws_capsule;

static PyObject *ws_add(py_workspace *self, PyObject *args, PyObject *kw) {
    static char *kwlist[] = { "source", "offset", "x", "length", "z", "tag", NULL };
    PyObject *src = Py_None, *tag = Py_None;
    long long offset = 0, x = 0, length = 0, z = 0;
    if (!PyArg_ParseTupleAndKeywords(args, kw, "|OLLLLO", kwlist, &src, &offset, &x, &length, &z, &tag)) return NULL;
    py_ws_item *it = (py_ws_item *)py_type_WorkspaceItem.tp_alloc(&py_type_WorkspaceItem, 0);
    if (!it) return NULL;
    if (!py_video_take_source(src, &it->source)) { Py_DECREF(it); return NULL; }
    Py_INCREF(tag);
    py_wrlock_nogil(&self->lock);
    it->item = workspace_add_item(self->ws, it->source, x, length, offset, z, tag);
    pthread_rwlock_unlock(&self->lock);
    it->owner = self;
    if (!it->item || PyList_Append(self->items, (PyObject *)it) < 0) { Py_DECREF(tag); Py_DECREF(it); return PyErr_NoMemory(); }
    return (PyObject *)it;
}

static PyObject *ws_remove(py_workspace *self, PyObject *args) {
    py_ws_item *it;
    if (!PyArg_ParseTuple(args, "O!", &py_type_WorkspaceItem, &it)) return NULL;
    if (it->owner != self) { PyErr_SetString(PyExc_ValueError, "The item does not belong to this workspace."); return NULL; }
    PyObject *tag = workspace_get_item_tag(it->item);
    py_wrlock_nogil(&self->lock);
    item_detach(it);
    pthread_rwlock_unlock(&self->lock);
    Py_XDECREF(tag);
    Py_ssize_t idx = PySequence_Index(self->items, (PyObject *)it);
    if (idx >= 0) PySequence_DelItem(self->items, idx); else PyErr_Clear();
    Py_RETURN_NONE;
}

static Py_ssize_t ws_len(py_workspace *self) { return workspace_get_length(self->ws); }

/* ws[i]: items in the library's order (by x, then z), as workspace_get_item gives them */
static PyObject *ws_item(py_workspace *self, Py_ssize_t i) {
    if (i < 0 || i >= ws_len(self)) { PyErr_SetString(PyExc_IndexError, "Index was out of range."); return NULL; }
    workspace_item_t *raw = workspace_get_item(self->ws, (int)i);
    for (Py_ssize_t k = 0; k < PyList_GET_SIZE(self->items); k++) {
        py_ws_item *it = (py_ws_item *)PyList_GET_ITEM(self->items, k);
        if (it->item == raw) { Py_INCREF(it); return (PyObject *)it; }
    }
    PyErr_SetString(PyExc_RuntimeError, "workspace item without a Python object");
    return NULL;
}

static PySequenceMethods ws_seq = { .sq_length = (lenfunc)ws_len, .sq_item = (ssizeargfunc)ws_item };
static PyMethodDef ws_methods[] = {
    { "add", (PyCFunction)ws_add, METH_VARARGS | METH_KEYWORDS, "add(source=, offset=, x=, length=, z=, tag=) -> item" },
    { "remove", (PyCFunction)ws_remove, METH_VARARGS, "remove(item)" },
    { NULL }
};
static PyGetSetDef ws_getset[] = { { VIDEO_FRAME_SOURCE_FUNCS, pyext_capsule_getter, NULL, "Video frame source C API.", &ws_capsule }, { NULL } };
static PyTypeObject py_type_Workspace = {
    PyVarObject_HEAD_INIT(NULL, 0)
    .tp_name = "fluggo.media.process.VideoWorkspace", .tp_basicsize = sizeof(py_workspace), .tp_flags = Py_TPFLAGS_DEFAULT,
    .tp_base = &py_type_VideoSource, .tp_new = PyType_GenericNew, .tp_init = (initproc)ws_init, .tp_dealloc = (destructor)ws_dealloc,
    .tp_getset = ws_getset, .tp_methods = ws_methods, .tp_as_sequence = &ws_seq,
};

/* ---- items */

static void item_dealloc(py_ws_item *self) {
    py_video_take_source(NULL, &self->source);
    Py_TYPE(self)->tp_free((PyObject *)self);
}
static bool item_alive(py_ws_item *self) {
    if (self->item) return true;
    PyErr_SetString(PyExc_RuntimeError, "The item has been removed from its workspace.");
    return false;
}
static PyObject *item_get_x(py_ws_item *self, void *c) { int64_t x; if (!item_alive(self)) return NULL; workspace_get_item_pos(self->item, &x, NULL, NULL); return PyLong_FromLongLong(x); }
static PyObject *item_get_length(py_ws_item *self, void *c) { int64_t v; if (!item_alive(self)) return NULL; workspace_get_item_pos(self->item, NULL, &v, NULL); return PyLong_FromLongLong(v); }
static PyObject *item_get_z(py_ws_item *self, void *c) { int64_t v; if (!item_alive(self)) return NULL; workspace_get_item_pos(self->item, NULL, NULL, &v); return PyLong_FromLongLong(v); }
static PyObject *item_get_offset(py_ws_item *self, void *c) { if (!item_alive(self)) return NULL; return PyLong_FromLongLong(workspace_get_item_offset(self->item)); }
static PyObject *item_get_source(py_ws_item *self, void *c) { PyObject *o = self->source ? (PyObject *)self->source->obj : Py_None; Py_INCREF(o); return o; }
static PyObject *item_get_tag(py_ws_item *self, void *c) { if (!item_alive(self)) return NULL; PyObject *t = workspace_get_item_tag(self->item); if (!t) t = Py_None; Py_INCREF(t); return t; }

static PyObject *item_update(py_ws_item *self, PyObject *args, PyObject *kw) {
    static char *kwlist[] = { "x", "length", "z", "offset", "source", "tag", NULL };
    PyObject *x = NULL, *length = NULL, *z = NULL, *offset = NULL, *source = NULL, *tag = NULL;
    if (!item_alive(self)) return NULL;
    if (!PyArg_ParseTupleAndKeywords(args, kw, "|OOOOOO", kwlist, &x, &length, &z, &offset, &source, &tag)) return NULL;
    int64_t vx, vl, vz, vo;
    if (x) vx = PyLong_AsLongLong(x);
    if (length) vl = PyLong_AsLongLong(length);
    if (z) vz = PyLong_AsLongLong(z);
    if (offset) vo = PyLong_AsLongLong(offset);
    if (PyErr_Occurred()) return NULL;
    video_source *fresh = NULL;
    if (source && !py_video_take_source(source, &fresh)) return NULL;
    void *src_ptr = fresh, *tag_ptr = tag;
    PyObject *old_tag = tag ? workspace_get_item_tag(self->item) : NULL;
    if (tag) Py_INCREF(tag);
    py_workspace *owner = self->owner;
    py_wrlock_nogil(&owner->lock);
    workspace_update_item(self->item, x ? &vx : NULL, length ? &vl : NULL, z ? &vz : NULL, offset ? &vo : NULL,
                          source ? &src_ptr : NULL, tag ? &tag_ptr : NULL);
    if (source) { video_source *old = self->source; self->source = fresh; py_video_take_source(NULL, &old); }
    pthread_rwlock_unlock(&owner->lock);
    Py_XDECREF(old_tag);
    Py_RETURN_NONE;
}
static PyGetSetDef item_getset[] = {
    { "x", (getter)item_get_x, NULL, "First frame of the item in workspace time." },
    { "length", (getter)item_get_length, NULL, "Length in frames." },
    { "z", (getter)item_get_z, NULL, "Stacking order; higher is on top." },
    { "offset", (getter)item_get_offset, NULL, "Source frame shown at x." },
    { "source", (getter)item_get_source, NULL, "The video source." },
    { "tag", (getter)item_get_tag, NULL, "User object." },
    { NULL }
};
static PyMethodDef item_methods[] = { { "update", (PyCFunction)item_update, METH_VARARGS | METH_KEYWORDS, "update(x=, length=, z=, offset=, source=, tag=)" }, { NULL } };
static PyTypeObject py_type_WorkspaceItem = {
    PyVarObject_HEAD_INIT(NULL, 0)
    .tp_name = "fluggo.media.process.VideoWorkspaceItem", .tp_basicsize = sizeof(py_ws_item), .tp_flags = Py_TPFLAGS_DEFAULT,
    .tp_dealloc = (destructor)item_dealloc, .tp_getset = item_getset, .tp_methods = item_methods,
};

/* ---------------------------------------------------------------- VideoPullQueue */

typedef struct pq_item {
    PyObject_HEAD
    struct pq_item *next;
    video_source *source;
    PyObject *callback, *user_data, *pyframe;
    PyObject *queue;                /* strong: a queue with requests pending stays alive (its last Python reference may be dropped
                                     * inside one of its own callbacks) */
    rgba_frame_f16 *frame;
    int frame_index;
    int owner;                      /* which of the queue's device contexts renders it: cvs_frame_owner(frame_index, ncontexts) */
    volatile int active;
} py_pq_item;

/* What the worker threads share.  It is NOT the Python object: the object can be deallocated on a worker thread (the last
 * reference dropped in a callback), and a worker must be able to go round its loop once more and leave.  The core is
 * reference-counted by the object and by every worker; the last one out frees it. */
#define PQ_MAX 16
typedef struct {
    pthread_mutex_t mutex;
    pthread_cond_t wake;
    py_pq_item *head[PQ_MAX], *tail[PQ_MAX];       /* one list per device context of the queue (one in all without `devices`) */
    int nctx;                                       /* lists in use */
    int ctx[PQ_MAX];                                /* the library's context id of each (-1: the default context) */
    int quit, refs;
} pq_core;

typedef struct { pq_core *core; int list; } pq_worker_arg;

typedef struct {
    PyObject_HEAD
    pq_core *core;
    pthread_t workers[PQ_MAX];
    int nworkers;
    int devices[PQ_MAX];                            /* HIP device of each context (as given), for the `devices` attribute */
} py_pullqueue;

static PyTypeObject py_type_PullQueueItem;

static void pq_core_unref(pq_core *c) {
    pthread_mutex_lock(&c->mutex);
    const int left = --c->refs;
    pthread_mutex_unlock(&c->mutex);
    if (left == 0) { pthread_cond_destroy(&c->wake); pthread_mutex_destroy(&c->mutex); free(c); }
}

static void *pq_worker(void *arg) {
    pq_core *q = ((pq_worker_arg *)arg)->core;
    const int me = ((pq_worker_arg *)arg)->list;                /* the list (= device context of the queue) this worker serves */
    free(arg);
    if (q->ctx[me] >= 0) cvs_set_context(q->ctx[me]);           /* every pull of this thread runs on that context's device */
    for (;;) {
        pthread_mutex_lock(&q->mutex);
        while (!q->head[me] && !q->quit) pthread_cond_wait(&q->wake, &q->mutex);
        if (!q->head[me] && q->quit) { pthread_mutex_unlock(&q->mutex); pq_core_unref(q); return NULL; }
        py_pq_item *it = q->head[me];
        q->head[me] = it->next;
        if (!q->head[me]) q->tail[me] = NULL;
        pthread_mutex_unlock(&q->mutex);

        if (it->active) {                                   /* the pull itself runs without the GIL (VideoPullQueue.c:99-105) */
            it->frame->current_window = it->frame->full_window;
            video_get_frame_f16(it->source, it->frame_index, it->frame);
        }
        PyGILState_STATE st = PyGILState_Ensure();
        if (it->active) {
            PyObject *r = PyObject_CallFunction(it->callback, "iOO", it->frame_index, it->pyframe, it->user_data);
            if (r) Py_DECREF(r); else PyErr_Print();
        }
        Py_CLEAR(it->callback); Py_CLEAR(it->user_data); Py_CLEAR(it->pyframe);
        py_video_take_source(NULL, &it->source);
        Py_CLEAR(it->queue);                                /* may run pq_dealloc right here, on this thread: only `q` is touched below */
        Py_DECREF(it);                                      /* the queue's reference */
        PyGILState_Release(st);
    }
}

/* VideoPullQueue(workers=2, devices=None): each worker thread pulls on its own HIP stream (the library binds one stream per
 * thread), so `workers` frames are in flight at a time; 2 is the reference's pool size (VideoPullQueue.c:110).
 * devices: a sequence of HIP device ordinals, e.g. range(8) on a node of eight GPUs.  The queue opens one device context of the
 * library per entry (cvs_context_open: its own scratch pool and table caches; naming a device twice gives two contexts on it),
 * binds worker w to context w % len(devices) and hands frame g to context g % len(devices) (cvs_frame_owner) -- the
 * frame-to-GPU rule of the whole build, inside one process.  The graph needs nothing done to it: its parameters are host
 * values, and every context builds the tables it needs on first use.  workers must be at least len(devices).  Without
 * `devices` every worker runs in the default context, as before. */
static int pq_init(py_pullqueue *self, PyObject *args, PyObject *kw) {
    static char *kwlist[] = { "workers", "devices", NULL };
    int workers = 2;
    PyObject *devices = NULL;
    if (!PyArg_ParseTupleAndKeywords(args, kw, "|iO", kwlist, &workers, &devices)) return -1;
    if (workers < 1 || workers > PQ_MAX) { PyErr_SetString(PyExc_ValueError, "workers must be between 1 and 16"); return -1; }
    if (self->core) { PyErr_SetString(PyExc_RuntimeError, "VideoPullQueue is already initialised"); return -1; }
    int ndev = 0, dev[PQ_MAX];
    if (devices && devices != Py_None) {
        PyObject *seq = PySequence_Fast(devices, "devices must be a sequence of HIP device ordinals");
        if (!seq) return -1;
        const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
        if (n < 1 || n > PQ_MAX || n > workers) { Py_DECREF(seq); PyErr_SetString(PyExc_ValueError, "devices: between 1 and `workers` entries (at most 16)"); return -1; }
        for (Py_ssize_t i = 0; i < n; i++) {
            const long d = PyLong_AsLong(PySequence_Fast_GET_ITEM(seq, i));
            if (d == -1 && PyErr_Occurred()) { Py_DECREF(seq); return -1; }
            dev[ndev++] = (int)d;
        }
        Py_DECREF(seq);
    }
    pq_core *c = calloc(1, sizeof *c);
    if (!c) { PyErr_NoMemory(); return -1; }
    c->nctx = 1;
    c->ctx[0] = -1;
    self->devices[0] = -1;
    if (ndev) {
        for (int i = 0; i < ndev; i++) {
            c->ctx[i] = cvs_context_open(dev[i]);
            self->devices[i] = dev[i];
            if (c->ctx[i] < 0) { free(c); PyErr_Format(PyExc_RuntimeError, "VideoPullQueue: no context on device %d: %s", dev[i], cvs_last_error()); return -1; }
        }
        c->nctx = ndev;
    }
    pthread_mutex_init(&c->mutex, NULL);
    pthread_cond_init(&c->wake, NULL);
    c->refs = 1;                                            /* the object's */
    self->core = c;
    for (int i = 0; i < workers; i++) {
        pq_worker_arg *a = malloc(sizeof *a);
        if (!a) break;
        a->core = c; a->list = i % c->nctx;
        pthread_mutex_lock(&c->mutex); c->refs++; pthread_mutex_unlock(&c->mutex);
        if (pthread_create(&self->workers[self->nworkers], NULL, pq_worker, a) == 0) self->nworkers++;
        else { free(a); pthread_mutex_lock(&c->mutex); c->refs--; pthread_mutex_unlock(&c->mutex); }
    }
    if (self->nworkers < c->nctx) { PyErr_SetString(PyExc_RuntimeError, "could not start a worker thread for every device"); return -1; }
    return 0;
}
/* .devices: the HIP device of each of the queue's contexts (an empty tuple: the default context); .contexts: their ids */
static PyObject *pq_get_devices(py_pullqueue *self, void *closure) {
    const pq_core *c = self->core;
    const int n = c && c->ctx[0] >= 0 ? c->nctx : 0;
    PyObject *t = PyTuple_New(n);
    for (int i = 0; t && i < n; i++) PyTuple_SET_ITEM(t, i, PyLong_FromLong(closure ? c->ctx[i] : self->devices[i]));
    return t;
}
static void pq_dealloc(py_pullqueue *self) {
    pq_core *c = self->core;
    if (c) {
        pthread_mutex_lock(&c->mutex);
        c->quit = 1;
        pthread_cond_broadcast(&c->wake);
        pthread_mutex_unlock(&c->mutex);
        const pthread_t me = pthread_self();
        Py_BEGIN_ALLOW_THREADS
        for (int i = 0; i < self->nworkers; i++) {
            if (pthread_equal(self->workers[i], me)) pthread_detach(self->workers[i]);     /* deallocated from inside a callback: that worker leaves by itself */
            else pthread_join(self->workers[i], NULL);
        }
        Py_END_ALLOW_THREADS
        pq_core_unref(c);
    }
    Py_TYPE(self)->tp_free((PyObject *)self);
}
static PyObject *pq_enqueue(py_pullqueue *self, PyObject *args, PyObject *kw) {
    static char *kwlist[] = { "source", "frame_index", "window", "callback", "user_data", NULL };
    PyObject *src, *window_obj, *callback, *user_data;
    int frame_index; box2i window;
    if (!PyArg_ParseTupleAndKeywords(args, kw, "OiOOO", kwlist, &src, &frame_index, &window_obj, &callback, &user_data)) return NULL;
    if (!py_parse_box2i(window_obj, &window)) return NULL;
    if (!self->core) { PyErr_SetString(PyExc_RuntimeError, "VideoPullQueue is not initialised"); return NULL; }
    py_pq_item *it = (py_pq_item *)py_type_PullQueueItem.tp_alloc(&py_type_PullQueueItem, 0);
    if (!it) return NULL;
    it->pyframe = py_RgbaFrameF16_new(&window, &it->frame);
    if (!it->pyframe || !py_video_take_source(src, &it->source)) { Py_DECREF(it); return NULL; }
    Py_INCREF(callback); Py_INCREF(user_data); Py_INCREF(self);
    it->callback = callback; it->user_data = user_data; it->queue = (PyObject *)self; it->frame_index = frame_index; it->active = 1;
    Py_INCREF(it);                                          /* one reference for the queue, one for the caller */
    pq_core *c = self->core;
    const int list = cvs_frame_owner(frame_index, c->nctx);
    it->owner = list;
    pthread_mutex_lock(&c->mutex);
    if (c->tail[list]) c->tail[list]->next = it; else c->head[list] = it;
    c->tail[list] = it;
    pthread_cond_broadcast(&c->wake);                       /* (the workers of one list are the ones that can take it) */
    pthread_mutex_unlock(&c->mutex);
    return (PyObject *)it;
}
static PyObject *pq_item_cancel(py_pq_item *self, PyObject *dummy) { self->active = 0; Py_RETURN_NONE; }
static void pq_item_dealloc(py_pq_item *self) {
    Py_CLEAR(self->callback); Py_CLEAR(self->user_data); Py_CLEAR(self->pyframe); Py_CLEAR(self->queue);
    py_video_take_source(NULL, &self->source);
    Py_TYPE(self)->tp_free((PyObject *)self);
}
static PyObject *pq_item_get_owner(py_pq_item *self, void *closure) { return PyLong_FromLong(self->owner); }
static PyGetSetDef pq_item_getset[] = {
    { "owner", (getter)pq_item_get_owner, NULL, "Index (into the queue's `devices`) of the device context that renders this frame: frame_index % len(devices).", NULL },
    { NULL }
};
static PyMethodDef pq_item_methods[] = { { "cancel", (PyCFunction)pq_item_cancel, METH_NOARGS, "Drop the request: the callback will not run." }, { NULL } };
static PyTypeObject py_type_PullQueueItem = {
    PyVarObject_HEAD_INIT(NULL, 0)
    .tp_name = "fluggo.media.process.VideoPullQueueItem", .tp_basicsize = sizeof(py_pq_item), .tp_flags = Py_TPFLAGS_DEFAULT,
    .tp_dealloc = (destructor)pq_item_dealloc, .tp_methods = pq_item_methods, .tp_getset = pq_item_getset,
};
static PyMethodDef pq_methods[] = {
    { "enqueue", (PyCFunction)pq_enqueue, METH_VARARGS | METH_KEYWORDS, "enqueue(source, frame_index, window, callback, user_data) -> item" },
    { NULL }
};
static PyGetSetDef pq_getset[] = {
    { "devices", (getter)pq_get_devices, NULL, "HIP device ordinal of each of the queue's device contexts (empty: the default context).", NULL },
    { "contexts", (getter)pq_get_devices, NULL, "The library's context id of each entry of `devices`.", (void *)1 },
    { NULL }
};
static PyTypeObject py_type_PullQueue = {
    PyVarObject_HEAD_INIT(NULL, 0)
    .tp_name = "fluggo.media.process.VideoPullQueue", .tp_basicsize = sizeof(py_pullqueue), .tp_flags = Py_TPFLAGS_DEFAULT,
    .tp_new = PyType_GenericNew, .tp_init = (initproc)pq_init, .tp_dealloc = (destructor)pq_dealloc, .tp_methods = pq_methods,
    .tp_getset = pq_getset,
};

int init_workspace(PyObject *module) {
    if (pyext_make_capsule(&ws_capsule, &ws_funcs) < 0) return -1;
    if (PyType_Ready(&py_type_WorkspaceItem) < 0 || PyType_Ready(&py_type_PullQueueItem) < 0) return -1;
    if (pyext_add_type(module, "VideoWorkspace", &py_type_Workspace) < 0) return -1;
    return pyext_add_type(module, "VideoPullQueue", &py_type_PullQueue);
}
