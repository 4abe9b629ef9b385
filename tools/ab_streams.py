#!/usr/bin/env python3
"""Is the chain kernel's run-to-run level a property of the STREAM (hardware queue) it is launched on?
One process, one ring of frames, eight streams measured in shuffled order, six rounds."""
import ctypes as C
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from canvas_amd import REC709_RGB_TO_YPBPR, _lib, synth  # noqa: E402
from canvas_amd.device import DeviceFrame, chain_color_over  # noqa: E402

lib = _lib.load()
_lib.check(lib.cvs_init(0))
lib.init_half()
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
streams = [lib.cvs_stream_create() for _ in range(8)]
arr = chain_color_over([jobs[i % 8] for i in range(64)], m, 0, -1, streams[0])
lib.cvs_stream_sync(streams[0])
e0, e1 = lib.cvs_event_create(), lib.cvs_event_create()


def run(s):
    ts = []
    for _ in range(4):
        lib.cvs_event_record(e0, s)
        lib.cvs_chain_color_over_f16_dev(arr, 64, mp, 0, -1, s)
        lib.cvs_event_record(e1, s)
        lib.cvs_stream_sync(s)
        ts.append(lib.cvs_event_elapsed_ms(e0, e1))
    return sorted(ts)[1]


for s in streams:
    run(s)
res = {i: [] for i in range(8)}
rng = random.Random(5)
for rep in range(6):
    order = list(range(8))
    rng.shuffle(order)
    for i in order:
        res[i].append(run(streams[i]))
for i in range(8):
    v = sorted(res[i])
    med = (v[2] + v[3]) / 2
    print("stream %d: median %.3f ms (min %.3f max %.3f) -> %.4f of 8 TB/s" % (i, med, v[0], v[-1], 64 * w * h * 24 / med / 1e6 / 8000))
