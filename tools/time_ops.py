#!/usr/bin/env python3
"""Per-entry-point timing of the device twins at 3840x2160 (HIP events on the launch stream): achieved GB/s against
each operation's algorithmic bytes.  Diagnostic: finds entry points that are far from the HBM rate."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from canvas_amd import REC709_RGB_TO_YPBPR, _lib, synth  # noqa: E402
from canvas_amd.abi import box2i  # noqa: E402
from canvas_amd.device import DeviceFrame  # noqa: E402

if os.environ.get("CANVAS_LIB"):                      # A/B runs: another build of the library (never set by the package)
    _lib.LIB_PATH = os.environ["CANVAS_LIB"]
lib = _lib.load()
_lib.check(lib.cvs_init(0))
lib.init_half()
stream = lib.cvs_stream_create()
w, h = 3840, 2160
full = (0, 0, w - 1, h - 1)
px = w * h
a16 = DeviceFrame.from_host(synth.layer_frame(w, h, 0, 0))
b16 = DeviceFrame.from_host(synth.layer_frame(w, h, 1, 0))
o16 = DeviceFrame(full, np.uint16)
a32, b32, o32 = DeviceFrame(full, np.float32), DeviceFrame(full, np.float32), DeviceFrame(full, np.float32)
lib.cvs_frame_f16_to_f32_dev(a32.ref(), a16.ref(), stream)
lib.cvs_frame_f16_to_f32_dev(b32.ref(), b16.ref(), stream)
m = np.array(REC709_RGB_TO_YPBPR, np.float32)
mp = m.ctypes.data_as(C.POINTER(C.c_float))
col = _lib.rgba_f32(0.25, 0.5, 0.75, 1.0)
win = box2i.of(*full)
bytes_out = lib.cvs_malloc(px * 4)


def reset32():
    a32.c.current_window = box2i.of(*full)


ops = [
    ("copy_frame_f16", 16, lambda: lib.cvs_copy_frame_f16_dev(o16.ref(), a16.ref(), stream)),
    ("frame_f16_to_f32", 24, lambda: lib.cvs_frame_f16_to_f32_dev(o32.ref(), a16.ref(), stream)),
    ("frame_f32_to_f16", 24, lambda: lib.cvs_frame_f32_to_f16_dev(o16.ref(), a32.ref(), stream)),
    ("copy_frame_alpha_f32(0.5)", 32, lambda: lib.cvs_copy_frame_alpha_f32_dev(o32.ref(), a32.ref(), C.c_float(0.5), stream)),
    ("mix_over_f32(1.0) in place", 48, lambda: (reset32(), lib.cvs_mix_over_f32_dev(a32.ref(), b32.ref(), C.c_float(1.0), stream))),
    ("mix_cross_f32(0.3)", 48, lambda: lib.cvs_mix_cross_f32_dev(o32.ref(), a32.ref(), b32.ref(), C.c_float(0.3), stream)),
    ("gain_offset_f16", 16, lambda: lib.cvs_gain_offset_f16_dev(o16.ref(), a16.ref(), C.c_float(1.5), C.c_float(0.0625), stream)),
    ("color_matrix_f16 in place (pre LUT)", 16, lambda: lib.cvs_color_matrix_f16_dev(b16.ref(), mp, 0, -1, stream)),
    ("color_matrix_f16_to (pre LUT)", 16, lambda: lib.cvs_color_matrix_f16_to_dev(o16.ref(), a16.ref(), mp, 0, -1, stream)),
    ("fill_solid_f16", 8, lambda: lib.cvs_fill_solid_f16_dev(o16.ref(), C.byref(win), C.byref(col), stream)),
    ("fill_solid_f32", 16, lambda: lib.cvs_fill_solid_f32_dev(o32.ref(), C.byref(win), C.byref(col), stream)),
    ("frame_to_bytes (sRGB, RGBA8)", 12, lambda: lib.cvs_frame_to_bytes_dev(bytes_out, a16.ref(), 3, 0, stream)),
    ("half_lookup (4 halfs/px)", 16, lambda: lib.cvs_half_lookup_dev(lib.cvs_lut_device(0), o16.ptr, a16.ptr, px * 4, stream)),
]
e0, e1 = lib.cvs_event_create(), lib.cvs_event_create()
print("%-40s %8s %9s %7s" % ("entry point (3840x2160)", "ms", "GB/s", "of 8T"))
for name, bpp, fn in ops:
    fn()
    lib.cvs_stream_sync(stream)
    ts = []
    for _ in range(7):
        lib.cvs_event_record(e0, stream)
        fn()
        lib.cvs_event_record(e1, stream)
        lib.cvs_stream_sync(stream)
        ts.append(lib.cvs_event_elapsed_ms(e0, e1))
    t = float(np.median(ts))
    gbs = px * bpp / t / 1e6
    print("%-40s %8.4f %9.0f %6.1f%%" % (name, t, gbs, gbs / 80))
