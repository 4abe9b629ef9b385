"""Launches of KNOWN size for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on access widths the guide calls
uncalibrated (MI355X_MICROARCH.md, HBM): 8 bytes per lane.  Each call reads / writes whole 3840x2160 frames, rotating
over 12 frame sets (well past the Infinity Cache):
  k_copy16        rgba_f16 -> rgba_f16   8 B/lane loads, 8 B/lane stores     66.36 MB read, 66.36 MB written per launch
  k_widen         rgba_f16 -> rgba_f32   8 B/lane loads, 16 B/lane stores    66.36 MB read, 132.71 MB written
  k_narrow        rgba_f32 -> rgba_f16   16 B/lane loads, 8 B/lane stores    132.71 MB read, 66.36 MB written
Run under  rocprofv3 --kernel-trace --pmc FETCH_SIZE  (and WRITE_SIZE); tools/summarize_r03.py derives the factors."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from canvas_amd import _lib, synth                      # noqa: E402
from canvas_amd.device import DeviceFrame               # noqa: E402

lib = _lib.load()
assert lib.cvs_init(0) == 0
lib.init_half()
w, h, n = 3840, 2160, 12
full = (0, 0, w - 1, h - 1)
host = synth.layer_frame(w, h, 1, 0)
h16 = [DeviceFrame.from_host(host)] + [DeviceFrame(full, np.uint16) for _ in range(n - 1)]
for d in h16[1:]:
    _lib.check(lib.cvs_memcpy_d2d(d.ptr, h16[0].ptr, d.nbytes, None))
o16 = [DeviceFrame(full, np.uint16) for _ in range(n)]
f32 = [DeviceFrame(full, np.float32) for _ in range(n)]
lib.cvs_stream_sync(None)
for rep in range(2):
    for i in range(n):
        _lib.check(lib.cvs_copy_frame_f16_dev(o16[i].ref(), h16[i].ref(), None))
    for i in range(n):
        _lib.check(lib.cvs_frame_f16_to_f32_dev(f32[i].ref(), h16[i].ref(), None))
    for i in range(n):
        _lib.check(lib.cvs_frame_f32_to_f16_dev(o16[i].ref(), f32[i].ref(), None))
lib.cvs_stream_sync(None)
print("calibration launches done: 2 x 12 each of copy16 / widen / narrow at %dx%d" % (w, h))
