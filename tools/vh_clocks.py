#!/usr/bin/env python3
"""Where a wave of k_fir_vh spends its life (diagnostic build only: make -C canvas_amd/csrc diag): the shader clock at seven points of
every workgroup of ONE 1920x1080 -> 3840x2160 f16 launch.   usage: python3 tools/vh_clocks.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools._diag import use_diag_library  # noqa: E402
use_diag_library()
from canvas_amd import _lib, synth  # noqa: E402
from canvas_amd.abi import v2f  # noqa: E402
from canvas_amd.device import DeviceFrame  # noqa: E402

lib = _lib.load()
_lib.check(lib.cvs_init(0))
lib.init_half()
w, h = 1920, 1080
srcs = [DeviceFrame.from_host(synth.layer_frame(w, h, 1, g)) for g in range(2)]
outs = [DeviceFrame((0, 0, 2 * w - 1, 2 * h - 1), np.uint16) for _ in range(24)]
NWG, SLOTS = 30 * 135 + 64, 8
buf = lib.cvs_malloc(NWG * SLOTS * 8)
lib.cvk_fir_vh_clock_buffer.restype, lib.cvk_fir_vh_clock_buffer.argtypes = C.c_int, [C.c_void_p]
for i in range(24):                                    # warm: tables, caches, clocks
    _lib.check(lib.cvs_scale_bilinear_f16_dev(outs[i].ref(), v2f(0, 0), srcs[i % 2].ref(), v2f(0, 0), v2f(2.0, 2.0), None))
_lib.check(lib.cvs_stream_sync(None))
_lib.check(lib.cvs_memset(buf, 0, NWG * SLOTS * 8, None))
_lib.check(lib.cvs_stream_sync(None))
assert lib.cvk_fir_vh_clock_buffer(buf) == 0
_lib.check(lib.cvs_scale_bilinear_f16_dev(outs[5].ref(), v2f(0, 0), srcs[1].ref(), v2f(0, 0), v2f(2.0, 2.0), None))
_lib.check(lib.cvs_stream_sync(None))
assert lib.cvk_fir_vh_clock_buffer(None) == 0
host = np.zeros((NWG, SLOTS), np.uint64)
_lib.check(lib.cvs_memcpy_d2h(host.ctypes.data, buf, host.nbytes, None))
t = host[host[:, 0] > 0].astype(np.int64)
t0 = t[:, 0].min()
names = ["start", "tap lists landed", "row range known", "line loop starts", "1st store issued", "8th store issued", "last store issued"]
print("%d workgroups; shader-clock ticks (s_memtime, 100 MHz constant clock on this part: 1 tick = 10 ns)" % len(t))
print("launch span (first start -> last end): %d ticks" % (t[:, 6].max() - t0))
print("%-20s %10s %10s %10s %10s" % ("point", "median", "p10", "p90", "max   (ticks after the launch's first start)"))
for k, n in enumerate(names):
    v = t[:, k] - t0
    print("%-20s %10d %10d %10d %10d" % (n, np.median(v), np.percentile(v, 10), np.percentile(v, 90), v.max()))
print("per workgroup, medians of the differences:")
for a, b in ((0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (0, 6)):
    d = t[:, b] - t[:, a]
    print("  %-18s -> %-18s %8d ticks (p10 %d, p90 %d)" % (names[a], names[b], np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
