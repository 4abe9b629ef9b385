#!/usr/bin/env python3
"""The blur node with 0..3 layers blended over it in the same launch (3840x2160 f16, 9 taps): what the epilogue costs.
Ten launches between two events, median of 7.   usage: time_blur_over.py [ntaps [columns]]   columns: 1 or 2 pins the
register-window blur to its one- or two-columns-per-lane form (cvs_fir_path_override), 0 leaves the choice to the library."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from canvas_amd import _lib, synth  # noqa: E402
from canvas_amd.device import DeviceFrame  # noqa: E402

if os.environ.get("CANVAS_DIAG") == "1":                   # the diagnostic build: CVS_BLUR_WIDTH / CVS_BLUR_ROWS / CVS_BLUR_WGS_PER_CU are read there only
    from tools._diag import use_diag_library
    use_diag_library()
if os.environ.get("CANVAS_LIB"):                      # A/B runs: another build of the library (never set by the package)
    _lib.LIB_PATH = os.environ["CANVAS_LIB"]
lib = _lib.load()
_lib.check(lib.cvs_init(0))
print("library %s, arithmetic %s" % (os.path.basename(_lib.LIB_PATH), "contracted" if lib.cvs_get_arithmetic() else "separate"), flush=True)
lib.init_half()
stream = lib.cvs_stream_create()
w, h = 3840, 2160
full = (0, 0, w - 1, h - 1)
src = DeviceFrame.from_host(synth.layer_frame(w, h, 0, 0))
over = [DeviceFrame.from_host(synth.layer_frame(w, h, k, 0)) for k in (1, 2, 3)]
out = DeviceFrame(full, np.uint16)
NT = int(sys.argv[1]) if len(sys.argv) > 1 else 9
COLS = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lib.cvs_fir_path_override({0: _lib.FIR_PATH_AUTO, 1: _lib.FIR_PATH_ONE_COLUMN, 2: _lib.FIR_PATH_TWO_COLUMNS}[COLS])
taps = synth.gaussian_taps(NT, 1.5)
tp = taps.ctypes.data_as(C.POINTER(C.c_float))
NAMES = {_lib.FIR_KERNEL_WINDOW: "k_blur", _lib.FIR_KERNEL_WINDOW_PAIR: "k_blur_pair"}
e0, e1 = lib.cvs_event_create(), lib.cvs_event_create()
for n in range(4):
    refs = (C.POINTER(_lib.rgba_frame_f16_t) * max(n, 1))(*[C.pointer(o.c) for o in over[:max(n, 1)]])
    fn = lambda: _lib.check(lib.cvs_blur_over_f16_dev(out.ref(), src.ref(), tp, NT, refs, n, stream))
    fn()
    lib.cvs_stream_sync(stream)
    ts = []
    for _ in range(7):
        lib.cvs_event_record(e0, stream)
        for _ in range(10):
            fn()
        lib.cvs_event_record(e1, stream)
        lib.cvs_stream_sync(stream)
        ts.append(lib.cvs_event_elapsed_ms(e0, e1) / 10)
    ms = sorted(ts)[3]
    bpp = 16 + 8 * n
    print("%d taps, blur + %d over: %.4f ms, %d B/px -> %.0f GB/s  (%s)" % (NT, n, ms, bpp, w * h * bpp / ms / 1e6, NAMES.get(lib.cvs_fir_last_kernel(), "?")))
