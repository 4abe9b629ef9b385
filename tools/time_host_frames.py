#!/usr/bin/env python3
"""BASELINE config 2 for a caller that owns HOST buffers: two 3840x2160 f16 layers go up, the fused chain runs, the
output comes back -- the PCIe-inclusive rate that DESIGN.md quotes beside (never instead of) the HBM-resident one.
Pageable numpy memory, one stream, then two streams with the transfers of one frame under the kernel of another."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from canvas_amd import REC709_RGB_TO_YPBPR, _lib, synth  # noqa: E402
from canvas_amd.device import DeviceFrame, chain_color_over  # noqa: E402

lib = _lib.load()
_lib.check(lib.cvs_init(0))
lib.init_half()
w, h = 3840, 2160
full = (0, 0, w - 1, h - 1)
m = np.array(REC709_RGB_TO_YPBPR, np.float32)
host_layers = [synth.layer_frame(w, h, k, 0).array for k in range(2)]
host_out = np.zeros((h, w, 4), np.uint16)
nbytes = host_out.nbytes


def lane():
    return lib.cvs_stream_create(), [DeviceFrame(full, np.uint16) for _ in range(2)], DeviceFrame(full, np.uint16)


def frame(lane_):
    s, layers, out = lane_
    for k in range(2):
        _lib.check(lib.cvs_memcpy_h2d(layers[k].ptr, host_layers[k].ctypes.data, nbytes, s))
    chain_color_over([(out, layers)], m, _lib.LUT_REC709_TO_LINEAR_SCENE, _lib.LUT_NONE, s)
    _lib.check(lib.cvs_memcpy_d2h(host_out.ctypes.data, out.ptr, nbytes, s))


for nlanes in (1, 2):
    lanes = [lane() for _ in range(nlanes)]
    for l in lanes:
        frame(l)
    for l in lanes:
        lib.cvs_stream_sync(l[0])
    n = 20
    t0 = time.perf_counter()
    for i in range(n):
        frame(lanes[i % nlanes])
    for l in lanes:
        lib.cvs_stream_sync(l[0])
    dt = time.perf_counter() - t0
    print("host frames, %d stream(s): %.2f ms per 4K frame = %.0f Mpx/s (%.1f GB/s over the link, 3 x %.1f MB per frame)"
          % (nlanes, dt / n * 1e3, n * w * h / dt / 1e6, 3 * nbytes * n / dt / 1e9, nbytes / 1e6))
