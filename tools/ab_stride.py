#!/usr/bin/env python3
"""Where the frames of a job lie relative to each other (and to the start of the allocation) against the chain kernel's
throughput, all layouts in ONE allocation of ONE process so that the luck of the allocation's own placement is the same
for every row.  3840x2160, 2 layers + output per job, 8 jobs cycled over 64-frame launches."""
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
arena_bytes = 6 << 30
arena = lib.cvs_malloc(arena_bytes)
print("arena at %#x" % arena)
# fill the arena with plausible pixels once (a 4K layer repeated): timing does not depend on the values' placement
px = synth.layer_pixels(w, h, 1, 0)
for off in range(0, arena_bytes - px.nbytes, 64 * MiB):
    _lib.check(lib.cvs_memcpy_h2d(arena + off, px.ctypes.data, px.nbytes, None))
lib.cvs_stream_sync(None)
e0, e1 = lib.cvs_event_create(), lib.cvs_event_create()


def measure_old(base_off, stride, label):
    jobs = []
    at = arena + base_off
    for g in range(8):
        frames = []
        for k in range(3):
            assert at + w * h * 8 <= arena + arena_bytes
            frames.append(DeviceFrame(full, np.uint16, ptr=at))
            at += stride
        jobs.append((frames[2], frames[:2]))
    batch = [jobs[i % 8] for i in range(64)]
    arr = chain_color_over(batch, m, _lib.LUT_REC709_TO_LINEAR_SCENE, _lib.LUT_NONE, stream)
    lib.cvs_stream_sync(stream)
    ts = []
    for _ in range(5):
        lib.cvs_event_record(e0, stream)
        lib.cvs_chain_color_over_f16_dev(arr, 64, m.ctypes.data_as(C.POINTER(C.c_float)), _lib.LUT_REC709_TO_LINEAR_SCENE, _lib.LUT_NONE, stream)
        lib.cvs_event_record(e1, stream)
        lib.cvs_stream_sync(stream)
        ts.append(lib.cvs_event_elapsed_ms(e0, e1))
    ms = sorted(ts)[2]
    print("%-44s %.3f ms  %.4f of 8 TB/s" % (label, ms, 64 * w * h * 24 / ms / 1e6 / 8000.0))


import ctypes as C  # noqa: E402
import random  # noqa: E402


def measure_ms(base_off, stride):
    jobs, at = [], arena + base_off
    for g in range(8):
        frames = []
        for k in range(3):
            assert at + w * h * 8 <= arena + arena_bytes
            frames.append(DeviceFrame(full, np.uint16, ptr=at))
            at += stride
        jobs.append((frames[2], frames[:2]))
    arr = chain_color_over([jobs[i % 8] for i in range(64)], m, _lib.LUT_REC709_TO_LINEAR_SCENE, _lib.LUT_NONE, stream)
    ts = []
    for _ in range(3):
        lib.cvs_event_record(e0, stream)
        lib.cvs_chain_color_over_f16_dev(arr, 64, m.ctypes.data_as(C.POINTER(C.c_float)), _lib.LUT_REC709_TO_LINEAR_SCENE, _lib.LUT_NONE, stream)
        lib.cvs_event_record(e1, stream)
        lib.cvs_stream_sync(stream)
        ts.append(lib.cvs_event_elapsed_ms(e0, e1))
    return sorted(ts)[1]


for _ in range(4):
    measure_ms(0, 64 * MiB)            # clocks up
configs = [(b * MiB, 64 * MiB) for b in range(0, 64, 2)] + [(0, s * MiB) for s in (66, 68, 72, 80, 96, 128, 130, 192)]
results = {c: [] for c in configs}
rng = random.Random(1)
for rep in range(4):
    order = configs[:]
    rng.shuffle(order)
    for c in order:
        results[c].append(measure_ms(*c))
for c in configs:
    ms = sorted(results[c])
    med = (ms[1] + ms[2]) / 2
    print("base +%2d MiB  stride %3d MiB   %.3f ms (min %.3f max %.3f)  %.4f of 8 TB/s" % (c[0] // MiB, c[1] // MiB, med, ms[0], ms[-1], 64 * w * h * 24 / med / 1e6 / 8000.0))
