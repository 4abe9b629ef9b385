"""ctypes mirror of the C-ABI structs in include/canvas_hip.h.

Layouts follow the reference's include/framework.h:46-75 (rational, v2i, box2i, v2f, box2f),
:155-177 (pixels, frames), :185-194,210-213 (source vtable), :618-627 (fir_filter).
Windows are inclusive int32 boxes; the canonical empty box is (0,0,-1,-1) (framework.h:96-102).
"""
import ctypes as C

import numpy as np


class v2i(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32)]


class box2i(C.Structure):
    _fields_ = [("min", v2i), ("max", v2i)]

    @classmethod
    def of(cls, x0, y0, x1, y1):
        return cls(v2i(x0, y0), v2i(x1, y1))

    @classmethod
    def empty(cls):
        return cls.of(0, 0, -1, -1)

    def tuple(self):
        return (self.min.x, self.min.y, self.max.x, self.max.y)

    def is_empty(self):
        return self.max.x < self.min.x or self.max.y < self.min.y

    @property
    def width(self):
        return 0 if self.max.x < self.min.x else self.max.x - self.min.x + 1

    @property
    def height(self):
        return 0 if self.max.y < self.min.y else self.max.y - self.min.y + 1

    def __repr__(self):
        return "box2i%r" % (self.tuple(),)


class v2f(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class rational(C.Structure):
    _fields_ = [("n", C.c_int32), ("d", C.c_uint32)]


class rgba_frame_f16(C.Structure):
    _fields_ = [("data", C.c_void_p), ("full_window", box2i), ("current_window", box2i)]


class rgba_frame_f32(C.Structure):
    _fields_ = [("data", C.c_void_p), ("full_window", box2i), ("current_window", box2i)]


class fir_filter(C.Structure):
    _fields_ = [("coeff", C.POINTER(C.c_float)), ("width", C.c_int), ("center", C.c_int)]


GET_FRAME_F16 = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.POINTER(rgba_frame_f16))
GET_FRAME_F32 = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.POINTER(rgba_frame_f32))


class video_frame_source_funcs(C.Structure):
    _fields_ = [("flags", C.c_int), ("get_frame", GET_FRAME_F16), ("get_frame_32", GET_FRAME_F32),
                ("get_frame_dev", C.c_void_p)]


class video_source(C.Structure):
    _fields_ = [("obj", C.c_void_p), ("funcs", C.POINTER(video_frame_source_funcs))]


class HostFrame:
    """A frame whose pixels live in a numpy array shaped (H, W, 4): uint16 for f16, float32 for f32."""

    def __init__(self, full_window, dtype, array=None, current_window=None, fill=None):
        fw = full_window if isinstance(full_window, box2i) else box2i.of(*full_window)
        self.dtype = np.dtype(dtype)
        assert self.dtype in (np.dtype(np.uint16), np.dtype(np.float32))
        h, w = fw.height, fw.width
        if array is None:
            array = np.zeros((h, w, 4), self.dtype) if fill is None else np.full((h, w, 4), fill, self.dtype)
        array = np.ascontiguousarray(array, self.dtype).reshape(h, w, 4)
        self.array = array
        cls = rgba_frame_f16 if self.dtype == np.uint16 else rgba_frame_f32
        cw = fw if current_window is None else (
            current_window if isinstance(current_window, box2i) else box2i.of(*current_window))
        self.c = cls(array.ctypes.data, box2i.of(*fw.tuple()), box2i.of(*cw.tuple()))

    @property
    def full_window(self):
        return self.c.full_window

    @property
    def current_window(self):
        return self.c.current_window

    def ref(self):
        return C.byref(self.c)

    def window_view(self, win=None):
        """numpy view of the pixels inside `win` (default: current_window)."""
        win = self.c.current_window if win is None else win
        if win.is_empty():
            return self.array[0:0, 0:0]
        fw = self.c.full_window
        return self.array[win.min.y - fw.min.y: win.max.y - fw.min.y + 1,
                          win.min.x - fw.min.x: win.max.x - fw.min.x + 1]

    def copy(self):
        return HostFrame(self.c.full_window, self.dtype, self.array.copy(), self.c.current_window)
