#!/usr/bin/env python3
"""Does exact black (r = g = b = 0: letterbox bars, titles on black) cost the fused chain anything?  Zero numerators
send a wave from the shared-reciprocal divide to three IEEE divides per pixel.  Measured: no (the launch stays
memory-bound; 0.57-0.67 ms per 16 frames whatever the fraction).
16 4K frames per launch, layer 1 translucent; a fraction of the rows painted black in both layers."""
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
w, h, n = 3840, 2160, 16
full = (0, 0, w - 1, h - 1)
m = np.array(REC709_RGB_TO_YPBPR, np.float32)
e0, e1 = lib.cvs_event_create(), lib.cvs_event_create()
for frac in (0.0, 0.1, 0.5, 1.0):
    jobs = []
    for g in range(n):
        base = DeviceFrame(full, np.uint16)
        base_px = synth.layer_pixels(w, h, 0, g)
        base_px[: int(h * frac), :, :3] = 0                  # black in every layer: only then are the numerators zero
        base.upload(base_px)
        top_px = synth.layer_pixels(w, h, 1, g)
        top_px[: int(h * frac), :, :3] = 0
        top = DeviceFrame(full, np.uint16)
        top.upload(top_px)
        jobs.append((DeviceFrame(full, np.uint16), [base, top]))
    for name, mat, pre in (("graded", m, _lib.LUT_REC709_TO_LINEAR_SCENE), ("plain stack", None, _lib.LUT_NONE)):
        arr = chain_color_over(jobs, mat, pre, _lib.LUT_NONE, stream)
        lib.cvs_stream_sync(stream)
        ts = []
        for _ in range(7):
            lib.cvs_event_record(e0, stream)
            chain_color_over(jobs, mat, pre, _lib.LUT_NONE, stream)
            lib.cvs_event_record(e1, stream)
            lib.cvs_stream_sync(stream)
            ts.append(lib.cvs_event_elapsed_ms(e0, e1))
        ms = sorted(ts)[3]
        print("black rows %3d %%  %-12s %.3f ms per %d frames = %.0f GB/s" % (frac * 100, name, ms, n, n * w * h * 24 / ms / 1e6))
    for out, layers in jobs:
        out.free()
        for l in layers:
            l.free()
