/*
 * pydv.c -- coded-image sources and the two DV 4:1:1 nodes.
 *
 *   CodedImageSource, CodedImage, py_coded_image_take_source ... src/process/CodedImageSource.c:28-272,
 *                                                                include/pyframework.h:121-132
 *   DVReconstructionFilter(source)  coded images -> VideoSource   src/process/DVReconstructionFilter.c:31-110
 *   DVSubsampleFilter(source)       VideoSource -> coded images   src/process/DVSubsampleFilter.c:31-110
 *
 * A coded image is a handful of byte planes in host memory (it is what a decoder hands over or an encoder
 * takes), so these two nodes are where bytes cross PCIe: the reconstruction node uploads three planes
 * (518 400 bytes) and renders straight into the device frame it was given; the subsample node pulls its
 * source into a device frame, converts there and downloads the three planes.
 */
#include "pyext.h"

static PyObject *coded_image_tuple;          /* collections.namedtuple("CodedImage", "data stride line_count") */

CVS_EXPORT bool py_coded_image_take_source(PyObject *source, CodedImageSourceHolder *holder) {
    PyObject *old = holder->source.obj;
    holder->source.obj = NULL;
    holder->source.funcs = NULL;
    Py_XDECREF(old);
    Py_CLEAR(holder->csource);
    if (source == NULL || source == Py_None) return true;
    PyObject *capsule = PyObject_GetAttrString(source, CODED_IMAGE_SOURCE_FUNCS);
    if (!capsule || !PyCapsule_IsValid(capsule, CODED_IMAGE_SOURCE_FUNCS)) {
        Py_XDECREF(capsule);
        PyErr_SetString(PyExc_Exception, "The source didn't have an acceptable " CODED_IMAGE_SOURCE_FUNCS " attribute.");
        return false;
    }
    Py_INCREF(source);
    holder->source.obj = source;
    holder->source.funcs = PyCapsule_GetPointer(capsule, CODED_IMAGE_SOURCE_FUNCS);
    holder->csource = capsule;
    return true;
}

/* ---------------------------------------------------------------- CodedImageSource (base, subclassable in Python) */

static PyObject *cis_get_frame(PyObject *self, PyObject *args) {
    int frame;
    if (!PyArg_ParseTuple(args, "i", &frame)) return NULL;
    CodedImageSourceHolder holder = { { 0 } };
    if (!py_coded_image_take_source(self, &holder)) return NULL;
    coded_image *image = NULL;
    if (holder.source.funcs && holder.source.funcs->getFrame) {
        Py_BEGIN_ALLOW_THREADS
        image = holder.source.funcs->getFrame(holder.source.obj, frame, 0);
        Py_END_ALLOW_THREADS
    }
    py_coded_image_take_source(NULL, &holder);
    if (!image) Py_RETURN_NONE;
    int count = 0;
    while (count < CODED_IMAGE_MAX_PLANES && image->data[count]) count++;
    PyObject *result = count ? PyList_New(count) : NULL;
    for (int i = 0; result && i < count; i++) {
        PyObject *bytes = PyByteArray_FromStringAndSize(image->data[i], (Py_ssize_t)image->stride[i] * image->line_count[i]);
        PyObject *member = bytes ? PyObject_CallFunction(coded_image_tuple, "Oii", bytes, image->stride[i], image->line_count[i]) : NULL;
        Py_XDECREF(bytes);
        if (!member) { Py_CLEAR(result); break; }
        PyList_SET_ITEM(result, i, member);
    }
    if (image->free_func) image->free_func(image);
    if (!result && !PyErr_Occurred()) Py_RETURN_NONE;
    return result;
}

/* vtable entry of the base type: ask the Python object (a subclass that overrides get_frame) */
static coded_image *cis_from_python(PyObject *self, int frame, int quality) {
    PyGILState_STATE g = PyGILState_Ensure();
    coded_image *image = NULL;
    PyObject *own = PyObject_GetAttrString((PyObject *)Py_TYPE(self), "get_frame");
    PyObject *base = PyObject_GetAttrString((PyObject *)&py_type_CodedImageSource, "get_frame");
    const bool overridden = own && base && own != base;
    Py_XDECREF(own); Py_XDECREF(base);
    PyObject *planes = overridden ? PyObject_CallMethod(self, "get_frame", "i", frame) : NULL;
    if (planes && planes != Py_None) {
        Py_ssize_t n = PySequence_Length(planes);
        int strides[CODED_IMAGE_MAX_PLANES] = { 0 }, lines[CODED_IMAGE_MAX_PLANES] = { 0 };
        Py_buffer views[CODED_IMAGE_MAX_PLANES];
        bool have[CODED_IMAGE_MAX_PLANES] = { false }, ok = n >= 0;
        if (n > CODED_IMAGE_MAX_PLANES) n = CODED_IMAGE_MAX_PLANES;
        for (Py_ssize_t p = 0; ok && p < n; p++) {
            PyObject *plane = PySequence_GetItem(planes, p);
            PyObject *data = plane ? PyObject_GetAttrString(plane, "data") : NULL;
            PyObject *stride = plane ? PyObject_GetAttrString(plane, "stride") : NULL;
            PyObject *count = plane ? PyObject_GetAttrString(plane, "line_count") : NULL;
            ok = data && stride && count;
            if (ok && data != Py_None) {
                strides[p] = (int)PyLong_AsLong(stride);
                lines[p] = (int)PyLong_AsLong(count);
                ok = !PyErr_Occurred() && strides[p] >= 0 && lines[p] >= 0 && PyObject_GetBuffer(data, &views[p], PyBUF_SIMPLE) == 0;
                if (ok) {
                    have[p] = true;
                    if (views[p].len != (Py_ssize_t)strides[p] * lines[p]) {
                        fprintf(stderr, "fluggo.media.process.CodedImageSource: plane %zd: expected %zd bytes, got %zd bytes.\n",
                                p, (Py_ssize_t)strides[p] * lines[p], views[p].len);
                        ok = false;
                    }
                }
            }
            Py_XDECREF(data); Py_XDECREF(stride); Py_XDECREF(count); Py_XDECREF(plane);
        }
        if (ok) image = coded_image_alloc(strides, lines, (int)n);
        for (Py_ssize_t p = 0; p < n; p++) {
            if (image && have[p] && image->data[p]) memcpy(image->data[p], views[p].buf, (size_t)views[p].len);
            if (have[p]) PyBuffer_Release(&views[p]);
        }
        if (!ok && image) { image->free_func(image); image = NULL; }
    }
    if (PyErr_Occurred()) PyErr_Print();
    Py_XDECREF(planes);
    PyGILState_Release(g);
    return image;
}

static coded_image_source_funcs cis_funcs = { 0, (coded_image_getFrameFunc)cis_from_python };
static PyObject *cis_capsule, *sub_capsule;
static PyMethodDef cis_methods[] = {
    { "get_frame", cis_get_frame, METH_VARARGS, "[CodedImage(data, stride, line_count), ...] or None = source.get_frame(frame)" },
    { NULL }
};
static PyGetSetDef cis_getset[] = { { CODED_IMAGE_SOURCE_FUNCS, pyext_capsule_getter, NULL, "Coded image source C API.", &cis_capsule }, { NULL } };

CVS_EXPORT PyTypeObject py_type_CodedImageSource = {
    PyVarObject_HEAD_INIT(NULL, 0)
    .tp_name = "fluggo.media.process.CodedImageSource", .tp_basicsize = sizeof(PyObject), .tp_new = PyType_GenericNew,
    .tp_flags = Py_TPFLAGS_DEFAULT | Py_TPFLAGS_BASETYPE, .tp_methods = cis_methods, .tp_getset = cis_getset,
};

/* ---------------------------------------------------------------- DVReconstructionFilter */

typedef struct { PyObject_HEAD CodedImageSourceHolder source; } py_dvrecon;

static int recon_init(py_dvrecon *self, PyObject *args, PyObject *kw) {
    static char *kwlist[] = { "source", NULL };
    PyObject *src;
    if (!PyArg_ParseTupleAndKeywords(args, kw, "O", kwlist, &src)) return -1;
    return py_coded_image_take_source(src, &self->source) ? 0 : -1;
}
static void recon_dealloc(py_dvrecon *self) {
    py_coded_image_take_source(NULL, &self->source);
    Py_TYPE(self)->tp_free((PyObject *)self);
}

static void recon_render(PyObject *o, int frame_index, rgba_frame_dev *f) {      /* native: f16 */
    py_dvrecon *self = (py_dvrecon *)o;
    box2i_set_empty(&f->current_window);
    if (!self->source.source.obj || !self->source.source.funcs || !self->source.source.funcs->getFrame) return;
    coded_image *image = self->source.source.funcs->getFrame(self->source.source.obj, frame_index, 0);
    if (!image) return;
    /* three planes up, in one pooled block */
    size_t off[3], total = 0;
    bool ok = image->data[0] && image->data[1] && image->data[2];
    for (int p = 0; ok && p < 3; p++) { off[p] = total; total += (((size_t)image->stride[p] * (size_t)image->line_count[p]) + 255) & ~(size_t)255; }
    char *block = ok ? cvs_pool_malloc(total ? total : 1, f->stream) : NULL;
    if (block) {
        coded_image dev = *image;
        for (int p = 0; ok && p < 3; p++) {
            dev.data[p] = block + off[p];
            ok = cvs_memcpy_h2d(dev.data[p], image->data[p], (size_t)image->stride[p] * (size_t)image->line_count[p], f->stream) == 0;
        }
        rgba_frame_f16 out = { f->data, f->full_window, f->full_window };
        if (ok && cvs_reconstruct_dv_dev(&out, &dev, f->stream) == 0) f->current_window = out.current_window;
        /* the host planes may be freed below: the uploads must have left them */
        cvs_stream_sync(f->stream);
        cvs_pool_free(block, f->stream);
    }
    if (image->free_func) image->free_func(image);
}
DEFINE_NODE_VTABLE(recon, CVS_FORMAT_F16, 1, 0)
static void *recon_unused[] __attribute__((unused)) = { (void *)recon_slot_32 };
static PyGetSetDef recon_getset[] = { { VIDEO_FRAME_SOURCE_FUNCS, pyext_capsule_getter, NULL, "Video frame source C API.", &recon_capsule }, { NULL } };
static PyTypeObject py_type_DVReconstructionFilter = {
    PyVarObject_HEAD_INIT(NULL, 0)
    .tp_name = "fluggo.media.process.DVReconstructionFilter", .tp_basicsize = sizeof(py_dvrecon), .tp_flags = Py_TPFLAGS_DEFAULT,
    .tp_base = &py_type_VideoSource, .tp_new = PyType_GenericNew, .tp_init = (initproc)recon_init,
    .tp_dealloc = (destructor)recon_dealloc, .tp_getset = recon_getset,
};

/* ---------------------------------------------------------------- DVSubsampleFilter */

typedef struct { PyObject_HEAD video_source *source; } py_dvsub;

static int sub_init(py_dvsub *self, PyObject *args, PyObject *kw) {
    static char *kwlist[] = { "source", NULL };
    PyObject *src;
    if (!PyArg_ParseTupleAndKeywords(args, kw, "O", kwlist, &src)) return -1;
    return py_video_take_source(src, &self->source) ? 0 : -1;
}
static void sub_dealloc(py_dvsub *self) {
    py_video_take_source(NULL, &self->source);
    Py_TYPE(self)->tp_free((PyObject *)self);
}

static coded_image *sub_get_frame(py_dvsub *self, int frame, int quality) {
    const int strides[3] = { 720, 180, 180 }, lines[3] = { 480, 480, 480 };
    box2i window;
    box2i_set(&window, 0, -1, 719, 478);                   /* DVSubsampleFilter.c:55-56 */
    coded_image *out = coded_image_alloc(strides, lines, 3);
    if (!out) return NULL;
    rgba_frame_dev d = { NULL, CVS_FORMAT_F16, window, window, NULL };
    size_t off[3], total = 0;
    for (int p = 0; p < 3; p++) { off[p] = total; total += (((size_t)strides[p] * 480) + 255) & ~(size_t)255; }
    d.data = cvs_pool_malloc(frame_bytes(&window, CVS_FORMAT_F16), NULL);
    char *block = cvs_pool_malloc(total, NULL);
    int rc = (d.data && block) ? 0 : -1;
    if (rc == 0) {
        video_get_frame_dev(self->source, frame, &d);
        coded_image dev = *out;
        for (int p = 0; p < 3; p++) dev.data[p] = block + off[p];
        rgba_frame_f16 in = { d.data, d.full_window, d.current_window };
        rc = cvs_subsample_dv_dev(&dev, &in, 0, NULL);     /* the pulled frame is scratch: no need to leave it encoded */
        for (int p = 0; rc == 0 && p < 3; p++) rc = cvs_memcpy_d2h(out->data[p], dev.data[p], (size_t)strides[p] * 480, NULL);
    }
    cvs_pool_free(block, NULL);
    cvs_pool_free(d.data, NULL);
    if (rc != 0) { out->free_func(out); return NULL; }
    return out;
}

static coded_image_source_funcs sub_funcs = { 0, (coded_image_getFrameFunc)sub_get_frame };
static PyGetSetDef sub_getset[] = { { CODED_IMAGE_SOURCE_FUNCS, pyext_capsule_getter, NULL, "Coded image source C API.", &sub_capsule }, { NULL } };
static PyTypeObject py_type_DVSubsampleFilter = {
    PyVarObject_HEAD_INIT(NULL, 0)
    .tp_name = "fluggo.media.process.DVSubsampleFilter", .tp_basicsize = sizeof(py_dvsub), .tp_flags = Py_TPFLAGS_DEFAULT,
    .tp_base = &py_type_CodedImageSource, .tp_new = PyType_GenericNew, .tp_init = (initproc)sub_init,
    .tp_dealloc = (destructor)sub_dealloc, .tp_getset = sub_getset,
};

int init_dv(PyObject *module) {
    PyObject *collections = PyImport_ImportModule("collections");
    if (!collections) return -1;
    coded_image_tuple = PyObject_CallMethod(collections, "namedtuple", "ss", "CodedImage", "data stride line_count");
    Py_DECREF(collections);
    if (!coded_image_tuple) return -1;
    PyObject_SetAttrString(coded_image_tuple, "__module__", PyUnicode_FromString("fluggo.media.process"));
    cis_capsule = PyCapsule_New(&cis_funcs, CODED_IMAGE_SOURCE_FUNCS, NULL);
    sub_capsule = PyCapsule_New(&sub_funcs, CODED_IMAGE_SOURCE_FUNCS, NULL);
    if (!cis_capsule || !sub_capsule || pyext_make_capsule(&recon_capsule, &recon_funcs) < 0) return -1;
    Py_INCREF(coded_image_tuple);
    if (PyModule_AddObject(module, "CodedImage", coded_image_tuple) < 0) return -1;
    if (pyext_add_type(module, "CodedImageSource", &py_type_CodedImageSource) < 0) return -1;
    if (pyext_add_type(module, "DVReconstructionFilter", &py_type_DVReconstructionFilter) < 0) return -1;
    return pyext_add_type(module, "DVSubsampleFilter", &py_type_DVSubsampleFilter);
}
