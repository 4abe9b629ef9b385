#!/usr/bin/env python3
"""What a write-dominated launch of one frame can reach on this chip: the ceiling the enlarging scaler (1 byte read per 4
written) should be held against.  Times, per launch and back to back on one stream (HIP events, median of 7 x 10 launches):
  hipMemsetAsync of a 4K / 8K f16 frame            (pure writes, the runtime's own kernel)
  cvs_fill_solid_f16_dev of a 4K / 8K frame        (pure writes, one pixel per lane)
  cvs_copy_frame_f16_dev 4K -> 4K                  (1 : 1)
  cvs_scale_bilinear_f16_dev 1080p -> 4K, 4K -> 8K (1 : 4)
Targets rotate over more than 1.5 GB so that nothing stays in the Infinity Cache.   usage: python3 tools/write_ceiling.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from canvas_amd import _lib, synth  # noqa: E402
from canvas_amd.abi import box2i, v2f  # noqa: E402
from canvas_amd.device import DeviceFrame  # noqa: E402

if os.environ.get("CANVAS_LIB"):
    _lib.LIB_PATH = os.environ["CANVAS_LIB"]
lib = _lib.load()
_lib.check(lib.cvs_init(0))
lib.init_half()
stream = lib.cvs_stream_create()
e0, e1 = lib.cvs_event_create(), lib.cvs_event_create()


def timed(fn, n):
    """fn(i) enqueues launch i; returns median ms per launch over 7 groups of n"""
    for i in range(n):
        fn(i)
    lib.cvs_stream_sync(stream)
    ts = []
    for _ in range(7):
        lib.cvs_event_record(e0, stream)
        for i in range(n):
            fn(i)
        lib.cvs_event_record(e1, stream)
        lib.cvs_stream_sync(stream)
        ts.append(lib.cvs_event_elapsed_ms(e0, e1) / n)
    return sorted(ts)[3]


def report(name, ms, read_b, written_b):
    tot = read_b + written_b
    print("%-44s %.4f ms  %6.1f MB read %6.1f MB written  %.2f TB/s (%.3f of 8)" % (name, ms, read_b / 1e6, written_b / 1e6, tot / ms / 1e9, tot / ms / 1e9 / 8), flush=True)


color = _lib.rgba_f32(0.25, 0.5, 0.75, 1.0)
for w, h, n in ((3840, 2160, 24), (7680, 4320, 6)):
    outs = [DeviceFrame((0, 0, w - 1, h - 1), np.uint16) for _ in range(n)]
    nb = outs[0].nbytes
    report("hipMemsetAsync %dx%d f16" % (w, h), timed(lambda i: lib.cvs_memset(outs[i % n].ptr, 0, nb, stream), n), 0, nb)
    win = box2i.of(0, 0, w - 1, h - 1)
    report("cvs_fill_solid_f16_dev %dx%d" % (w, h), timed(lambda i: lib.cvs_fill_solid_f16_dev(outs[i % n].ref(), C.byref(win), C.byref(color), stream), n), 0, nb)
    if w == 3840:
        srcs = [DeviceFrame.from_host(synth.layer_frame(w, h, 1, 0)) for _ in range(2)] + [DeviceFrame((0, 0, w - 1, h - 1), np.uint16) for _ in range(n - 2)]
        report("cvs_copy_frame_f16_dev %dx%d" % (w, h), timed(lambda i: lib.cvs_copy_frame_f16_dev(outs[i % n].ref(), srcs[i % n].ref(), stream), n), nb, nb)
        for d in srcs:
            d.free()
    sw, sh = w // 2, h // 2
    small = [DeviceFrame.from_host(synth.layer_frame(sw, sh, 1, g % 2)) for g in range(n)]
    report("cvs_scale_bilinear_f16_dev %dx%d -> %dx%d" % (sw, sh, w, h),
           timed(lambda i: lib.cvs_scale_bilinear_f16_dev(outs[i % n].ref(), v2f(0, 0), small[i % n].ref(), v2f(0, 0), v2f(2.0, 2.0), stream), n), small[0].nbytes, nb)
    for d in outs + small:
        d.free()
