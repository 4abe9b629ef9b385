/*
 * pyext.h -- internals of the fluggo.media.process extension module built on libcanvas_hip.so.
 *
 * Every node type (filter / source) renders into a DEVICE frame: its vtable carries
 * VIDEO_SOURCE_FLAG_DEVICE and a slot-3 entry, so a graph of these nodes stays in HBM from the
 * sources to the consumer.  The host slots the reference's vtable also has (get_frame /
 * get_frame_32) are thin edges over the same render: allocate a device frame, render, copy back once.
 */
#ifndef CANVAS_PYEXT_H
#define CANVAS_PYEXT_H

#define PY_SSIZE_T_CLEAN
#include "pyframework.h"
#include <pthread.h>

/* Node locks and the GIL.  Worker threads (pull queue, playback) hold the reader side of a node's rwlock across a pull,
 * and a pull may need the GIL (a source or a frame function written in Python).  Python-called methods take either side
 * with the GIL held.  The one rule that keeps the pair (lock, GIL) free of cycles: NOBODY BLOCKS ON A NODE LOCK WHILE
 * HOLDING THE GIL.  Both sides try first and, if the lock is busy, wait for it with the GIL released.  A thread may then
 * wait for the GIL while it holds the lock (the writer coming back from its wait, a worker's reader calling into
 * Python): whoever has the GIL at that moment either runs on or gives it up before it blocks on the lock, so the wait ends.
 * (The first form released the GIL only on the writer side: a writer coming back for the GIL with the lock in hand met
 * a reader that had blocked on the lock with the GIL in hand -- anim.add() against get_values() from a callback.) */
static inline void py_wrlock_nogil(pthread_rwlock_t *lock) {
    if (pthread_rwlock_trywrlock(lock) == 0) return;
    Py_BEGIN_ALLOW_THREADS
    pthread_rwlock_wrlock(lock);
    Py_END_ALLOW_THREADS
}
/* reader side: called from worker threads without the GIL and from Python-called code with it */
static inline void py_rdlock(pthread_rwlock_t *lock) {
    if (pthread_rwlock_tryrdlock(lock) == 0) return;
    if (PyGILState_Check()) {
        Py_BEGIN_ALLOW_THREADS
        pthread_rwlock_rdlock(lock);
        Py_END_ALLOW_THREADS
    } else {
        pthread_rwlock_rdlock(lock);
    }
}

/* a node's native render: fill `frame` (device, of the node's native format) for frame_index */
typedef void (*node_render_func)(PyObject *self, int frame_index, rgba_frame_dev *frame);

/* shared plumbing (pymodule.c) */
void node_get_frame_dev(PyObject *self, int frame_index, rgba_frame_dev *frame, int native_format, node_render_func render);
void node_get_frame_host16(PyObject *self, int frame_index, rgba_frame_f16 *frame, int native_format, node_render_func render);
void node_get_frame_host32(PyObject *self, int frame_index, rgba_frame_f32 *frame, int native_format, node_render_func render);
size_t frame_bytes(const box2i *full, int format);

/* one static vtable + capsule per type, exposed through the `_video_frame_source_funcs` attribute */
PyObject *pyext_capsule_getter(PyObject *self, void *closure);      /* closure = PyObject** (capsule slot) */
int pyext_add_type(PyObject *module, const char *name, PyTypeObject *type);
int pyext_make_capsule(PyObject **slot, video_frame_source_funcs *funcs);

/* frame objects (pyframes.c) */
PyObject *py_RgbaFrameF16_new(box2i *full_window, rgba_frame_f16 **frame);
PyObject *py_RgbaFrameF32_new(box2i *full_window, rgba_frame_f32 **frame);
PyObject *py_get_frame_f16(PyObject *self, PyObject *args, PyObject *kw);
PyObject *py_get_frame_f32(PyObject *self, PyObject *args, PyObject *kw);
PyObject *py_get_frame_argb32(PyObject *self, PyObject *args, PyObject *kw);
PyObject *py_get_frame_rgba8(PyObject *self, PyObject *args, PyObject *kw);

int init_frames(PyObject *module);
int init_framefuncs(PyObject *module);
int init_animation(PyObject *module);
int init_dv(PyObject *module);
int init_sources(PyObject *module);
int init_workspace(PyObject *module);

/* node vtable boilerplate: DEFINE_NODE_VTABLE(Prefix, CVS_FORMAT_F16 or _F32, host16?, host32?) */
#define DEFINE_NODE_VTABLE(P, NATIVE, HOST16, HOST32)                                                          \
    static void P##_slot_dev(PyObject *self, int i, rgba_frame_dev *f) { node_get_frame_dev(self, i, f, NATIVE, P##_render); }   \
    static void P##_slot_16(PyObject *self, int i, rgba_frame_f16 *f) { node_get_frame_host16(self, i, f, NATIVE, P##_render); } \
    static void P##_slot_32(PyObject *self, int i, rgba_frame_f32 *f) { node_get_frame_host32(self, i, f, NATIVE, P##_render); } \
    static video_frame_source_funcs P##_funcs = {                                                             \
        .flags = VIDEO_SOURCE_FLAG_DEVICE,                                                                    \
        .get_frame = (HOST16) ? (video_get_frame_func)P##_slot_16 : NULL,                                     \
        .get_frame_32 = (HOST32) ? (video_get_frame_32_func)P##_slot_32 : NULL,                               \
        .get_frame_dev = (video_get_frame_dev_func)P##_slot_dev };                                            \
    static PyObject *P##_capsule;

#endif
