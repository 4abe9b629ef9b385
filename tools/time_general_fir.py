#!/usr/bin/env python3
"""The general FIR paths at 3840x2160 (per-line tap tables, LDS tiles): Lanczos at factors that are not 1/2, long and
even-length blurs, the triangle scaler at odd factors.  ms per call and GB/s against source + target bytes."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from canvas_amd import _lib, synth  # noqa: E402
from canvas_amd.abi import v2f  # noqa: E402
from canvas_amd.device import DeviceFrame  # noqa: E402

if "diag" in sys.argv[1:]:                     # the diagnostic build: CVS_LANES_ROWS, CVS_LANES_SKIP (sweep_ops.hip)
    sys.argv.remove("diag")
    from _diag import use_diag_library
    use_diag_library()
lib = _lib.load()
_lib.check(lib.cvs_init(0))
lib.init_half()
stream = lib.cvs_stream_create()
w, h = 3840, 2160
src16 = DeviceFrame.from_host(synth.layer_frame(w, h, 1, 0))
src32 = DeviceFrame((0, 0, w - 1, h - 1), np.float32)
lib.cvs_frame_f16_to_f32_dev(src32.ref(), src16.ref(), stream)
e0, e1 = lib.cvs_event_create(), lib.cvs_event_create()


def timed(name, fn, nbytes):
    fn()
    lib.cvs_stream_sync(stream)
    ts = []
    for _ in range(5):
        lib.cvs_event_record(e0, stream)
        for _ in range(4):
            fn()
        lib.cvs_event_record(e1, stream)
        lib.cvs_stream_sync(stream)
        ts.append(lib.cvs_event_elapsed_ms(e0, e1) / 4)
    ms = sorted(ts)[2]
    print("%-58s %.3f ms  %6.0f GB/s" % (name, ms, nbytes / ms / 1e6))


which = {"hv": _lib.FIR_PATH_HV, "passes": _lib.FIR_PATH_PASSES, "tiled": _lib.FIR_PATH_TILED}
pin = [a for a in sys.argv[1:] if a in which]
only_f16 = "f16" in sys.argv[1:]                 # just the f16 -> f16 resampler (one kernel under the profiler)
factors = [float(a) for a in sys.argv[1:] if a not in which and a != "f16"] or [0.4, 0.75, 1.5]
if pin:                                       # one kernel of the table path pinned (blurs too go to the tables then)
    lib.cvs_fir_path_override(which[pin[0]] | _lib.FIR_PATH_TABLES)
    print("general FIR path pinned to:", pin[0])
for f in factors:
    tw, th = int(w * f), int(h * f)
    out16 = DeviceFrame((0, 0, tw - 1, th - 1), np.uint16)
    timed("1-tap blur + Lanczos3 f16 -> f16, factor %.2f (%dx%d)" % (f, tw, th),
          lambda: _lib.check(lib.cvs_resample_lanczos_f16_dev(out16.ref(), src16.ref(), C.c_float(f), C.c_float(f), 3, stream)) if hasattr(lib, "cvs_resample_lanczos_f16_dev")
          else _lib.check(lib.cvs_blur_lanczos_f16_dev(out16.ref(), src16.ref(), np.array([1.0], np.float32).ctypes.data_as(C.POINTER(C.c_float)), 1, C.c_float(f), C.c_float(f), 3, stream)),
          w * h * 8 + tw * th * 8)
    if only_f16:
        out16.free()
        continue
    out32 = DeviceFrame((0, 0, tw - 1, th - 1), np.float32)
    timed("Lanczos3 f32 -> f32, factor %.2f" % f,
          lambda: _lib.check(lib.cvs_resample_lanczos_f32_dev(out32.ref(), src32.ref(), C.c_float(f), C.c_float(f), 3, stream)), w * h * 16 + tw * th * 16)
    timed("triangle scaler f32 -> f32, factor %.2f" % f,
          lambda: _lib.check(lib.cvs_scale_bilinear_f32_dev(out32.ref(), v2f(0, 0), src32.ref(), v2f(0, 0), v2f(f, f), stream)), w * h * 16 + tw * th * 16)
    out16.free(); out32.free()
for ntaps in (() if only_f16 else (9, 10, 21, 31)):
    taps = synth.gaussian_taps(ntaps | 1, ntaps / 6.0)[:ntaps].copy()
    out16 = DeviceFrame((0, 0, w - 1, h - 1), np.uint16)
    timed("blur f16 -> f16, %d taps" % ntaps, lambda: _lib.check(lib.cvs_fir_blur_f16_dev(out16.ref(), src16.ref(), taps.ctypes.data_as(C.POINTER(C.c_float)), ntaps, stream)), w * h * 16)
    out16.free()
