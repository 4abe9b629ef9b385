#!/usr/bin/env python3
"""Is the chain kernel's throughput level a property of the PROCESS or of the ALLOCATION the frames live in?
One process; the ring of frames (8 jobs x 3 frames of 64 MiB slots) is allocated, measured and freed several times, with
other allocations made in between so that it lands somewhere else each time.  (Answer: of neither -- eight rings in
one process all run at that process's level.)"""
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
stream = lib.cvs_stream_create()
w, h = 3840, 2160
full = (0, 0, w - 1, h - 1)
MiB = 1 << 20
m = np.array(REC709_RGB_TO_YPBPR, np.float32)
mp = m.ctypes.data_as(C.POINTER(C.c_float))
px = synth.layer_pixels(w, h, 1, 0)
e0, e1 = lib.cvs_event_create(), lib.cvs_event_create()
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
kept = []
arenas = []
spacer_gib = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # > 0: that much device memory between consecutive rings
for trial in range(8 if not spacer_gib else 4):    # rings alive at the same time: different places for certain
    kept.append(lib.cvs_malloc((spacer_gib << 30) if spacer_gib else rng.choice([2, 6, 30, 62, 126, 254, 510, 1022]) * MiB))
    arenas.append(lib.cvs_malloc(24 * 64 * MiB))
for trial, arena in enumerate(arenas * 2):
    for k in range(24):
        _lib.check(lib.cvs_memcpy_h2d(arena + k * 64 * MiB, px.ctypes.data, px.nbytes, None))
    lib.cvs_stream_sync(None)
    jobs = []
    for g in range(8):
        f = [DeviceFrame(full, np.uint16, ptr=arena + (3 * g + k) * 64 * MiB) for k in range(3)]
        jobs.append((f[2], f[:2]))
    arr = chain_color_over([jobs[i % 8] for i in range(64)], m, _lib.LUT_REC709_TO_LINEAR_SCENE, _lib.LUT_NONE, stream)
    ts = []
    for _ in range(6):
        lib.cvs_event_record(e0, stream)
        lib.cvs_chain_color_over_f16_dev(arr, 64, mp, _lib.LUT_REC709_TO_LINEAR_SCENE, _lib.LUT_NONE, stream)
        lib.cvs_event_record(e1, stream)
        lib.cvs_stream_sync(stream)
        ts.append(lib.cvs_event_elapsed_ms(e0, e1))
    ms = sorted(ts[1:])[2]
    print("trial %d: ring at %#x  %.3f ms  %.4f of 8 TB/s" % (trial, arena, ms, 64 * w * h * 24 / ms / 1e6 / 8000.0))
