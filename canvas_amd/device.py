"""Device-resident frames for callers that drive the `cvs_*_dev` entry points from Python
(tests, bench.py, the frame-sharding driver).  Memory comes from the library's own allocator
(hipMalloc); nothing here needs torch.
"""
import ctypes as C

import numpy as np

from . import _lib
from .abi import HostFrame, box2i, rgba_frame_f16, rgba_frame_f32


class DeviceFrame:
    """An rgba_frame_f16/f32 whose `data` points into HBM."""

    def __init__(self, full_window, dtype, current_window=None, ptr=None):
        """ptr: place the frame at this device address (memory owned by the caller, e.g. one arena
        holding a whole ring of frames); default: own allocation."""
        lib = _lib.load()
        fw = full_window if isinstance(full_window, box2i) else box2i.of(*full_window)
        self.dtype = np.dtype(dtype)
        assert self.dtype in (np.dtype(np.uint16), np.dtype(np.float32))
        self.height, self.width = fw.height, fw.width
        self.nbytes = self.height * self.width * 4 * self.dtype.itemsize
        self.owns = ptr is None
        self.ptr = lib.cvs_malloc(max(self.nbytes, 1)) if ptr is None else ptr
        if not self.ptr:
            raise MemoryError("cvs_malloc(%d): %s" % (self.nbytes, _lib.last_error()))
        cls = rgba_frame_f16 if self.dtype == np.uint16 else rgba_frame_f32
        cw = fw if current_window is None else (
            current_window if isinstance(current_window, box2i) else box2i.of(*current_window))
        self.c = cls(self.ptr, box2i.of(*fw.tuple()), box2i.of(*cw.tuple()))

    @classmethod
    def from_host(cls, host: HostFrame, stream=None):
        f = cls(host.full_window, host.dtype, host.current_window)
        f.upload(host.array, stream)
        return f

    def upload(self, array, stream=None):
        a = np.ascontiguousarray(array, self.dtype)
        assert a.nbytes == self.nbytes, (a.nbytes, self.nbytes)
        _lib.check(_lib.load().cvs_memcpy_h2d(self.ptr, a.ctypes.data, self.nbytes, stream), "h2d")

    def download(self, stream=None) -> HostFrame:
        out = np.empty((self.height, self.width, 4), self.dtype)
        if self.nbytes:
            _lib.check(_lib.load().cvs_memcpy_d2h(out.ctypes.data, self.ptr, self.nbytes, stream), "d2h")
        return HostFrame(self.c.full_window, self.dtype, out, self.c.current_window)

    @property
    def full_window(self):
        return self.c.full_window

    @property
    def current_window(self):
        return self.c.current_window

    def ref(self):
        return C.byref(self.c)

    def free(self):
        if self.ptr:
            if self.owns:
                _lib.load().cvs_free(self.ptr)
            self.ptr = None
            self.c.data = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def chain_color_over(jobs, matrix, pre_lut=_lib.LUT_NONE, post_lut=_lib.LUT_NONE, stream=None):
    """jobs: list of (out DeviceFrame, [layer DeviceFrames bottom first]).  Enqueues on `stream`."""
    lib = _lib.load()
    arr = (_lib.chain_job * len(jobs))()
    for i, (out, layers) in enumerate(jobs):
        arr[i].out = C.pointer(out.c)
        for k, l in enumerate(layers):
            arr[i].layers[k] = C.pointer(l.c)
        arr[i].nlayers = len(layers)
    m = None if matrix is None else np.ascontiguousarray(matrix, np.float32).reshape(9)      # None: plain over stack
    rc = lib.cvs_chain_color_over_f16_dev(arr, len(jobs), None if m is None else m.ctypes.data_as(C.POINTER(C.c_float)), pre_lut, post_lut, stream)
    _lib.check(rc, "cvs_chain_color_over_f16_dev")
    return arr
