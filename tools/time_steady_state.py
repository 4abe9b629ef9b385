#!/usr/bin/env python3
"""Per-launch duration of the bench workload over a few seconds: does the level move once the clocks have settled?
64-frame launches of BASELINE config 2 on a ring of 8 frame sets, HIP events around every launch."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from canvas_amd import REC709_RGB_TO_YPBPR, _lib, synth  # noqa: E402
from canvas_amd.device import DeviceFrame, chain_color_over  # noqa: E402

lib = _lib.load()
_lib.check(lib.cvs_init(0))
lib.init_half()
stream = lib.cvs_stream_create()
w, h = 3840, 2160
full = (0, 0, w - 1, h - 1)
MiB = 1 << 20
m = np.array(REC709_RGB_TO_YPBPR, np.float32)
mp = m.ctypes.data_as(C.POINTER(C.c_float))
arena = lib.cvs_malloc(24 * 64 * MiB)
for k in range(24):
    px = synth.layer_pixels(w, h, k % 3 if k % 3 < 2 else 1, k // 3)
    _lib.check(lib.cvs_memcpy_h2d(arena + k * 64 * MiB, px.ctypes.data, px.nbytes, None))
jobs = []
for g in range(8):
    f = [DeviceFrame(full, np.uint16, ptr=arena + (3 * g + k) * 64 * MiB) for k in range(3)]
    jobs.append((f[2], f[:2]))
arr = chain_color_over([jobs[i % 8] for i in range(64)], m, _lib.LUT_REC709_TO_LINEAR_SCENE, _lib.LUT_NONE, stream)
lib.cvs_stream_sync(stream)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
ev = [lib.cvs_event_create() for _ in range(n + 1)]
lib.cvs_event_record(ev[0], stream)
for i in range(n):
    lib.cvs_chain_color_over_f16_dev(arr, 64, mp, _lib.LUT_REC709_TO_LINEAR_SCENE, _lib.LUT_NONE, stream)
    lib.cvs_event_record(ev[i + 1], stream)
lib.cvs_stream_sync(stream)
ms = np.array([lib.cvs_event_elapsed_ms(ev[i], ev[i + 1]) for i in range(n)])
frac = 64 * w * h * 24 / ms / 1e6 / 8000.0
print("ring at %#x" % arena)
for lo in range(0, n, n // 12):
    seg = frac[lo:lo + n // 12]
    print("launches %4d-%4d (t = %5.2f s): median %.4f  min %.4f  max %.4f of 8 TB/s" % (lo, lo + len(seg) - 1, ms[:lo].sum() / 1e3, np.median(seg), seg.min(), seg.max()))
