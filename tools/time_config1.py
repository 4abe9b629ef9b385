#!/usr/bin/env python3
"""BASELINE config 1 through the Python surface: 1920x1080 SolidColor -> gain 1.5 / offset 0.0625 -> pull 100 frames
(host frames: the D2H copy of every frame is included), plus the same pull through the preview edge."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluggo.media import process  # noqa: E402
from fluggo.media.basetypes import box2i  # noqa: E402

assert process.check_context_supported(), process.last_error()
graph = process.VideoGainOffsetFilter(process.SolidColorVideoSource((0.25, 0.5, 0.75, 1.0)), gain=1.5, offset=0.0625)
window = box2i(0, 0, 1919, 1079)
for name, pull in [("get_frame_f16", graph.get_frame_f16), ("get_frame_argb32", graph.get_frame_argb32)]:
    pull(0, window)
    t0 = time.perf_counter()
    for i in range(100):
        pull(i, window)
    dt = time.perf_counter() - t0
    print("%s: 100 frames in %.1f ms = %.0f Mpx/s" % (name, dt * 1e3, 100 * 1920 * 1080 / dt / 1e6))
