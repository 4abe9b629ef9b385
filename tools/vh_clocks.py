#!/usr/bin/env python3
"""Where a wave of k_fir_vh spends its life (diagnostic build only: make -C canvas_amd/csrc diag): the shader clock at seven points of
every workgroup of ONE 1920x1080 -> 3840x2160 f16 launch.   usage: python3 tools/vh_clocks.py [tiles]
With `tiles`: the same for k_fir_tile_vh (eight points).  The clock is s_memtime's constant 100 MHz one: 1 tick = 10 ns."""
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

TILES = len(sys.argv) > 1 and sys.argv[1] == "tiles"
lib = _lib.load()
_lib.check(lib.cvs_init(0))
if not TILES:
    lib.cvs_fir_path_override(_lib.FIR_PATH_STRIPS)
lib.init_half()
w, h = 1920, 1080
srcs = [DeviceFrame.from_host(synth.layer_frame(w, h, 1, g)) for g in range(2)]
outs = [DeviceFrame((0, 0, 2 * w - 1, 2 * h - 1), np.uint16) for _ in range(24)]
NWG, SLOTS = 30 * 136 + 64, 16 if TILES else 8
buf = lib.cvs_malloc(NWG * SLOTS * 8)
set_buffer = lib.cvk_fir_tvh_clock_buffer if TILES else lib.cvk_fir_vh_clock_buffer
set_buffer.restype, set_buffer.argtypes = C.c_int, [C.c_void_p]
for i in range(24):                                    # warm: tables, caches, clocks
    _lib.check(lib.cvs_scale_bilinear_f16_dev(outs[i].ref(), v2f(0, 0), srcs[i % 2].ref(), v2f(0, 0), v2f(2.0, 2.0), None))
_lib.check(lib.cvs_stream_sync(None))
_lib.check(lib.cvs_memset(buf, 0, NWG * SLOTS * 8, None))
_lib.check(lib.cvs_stream_sync(None))
assert set_buffer(buf) == 0
_lib.check(lib.cvs_scale_bilinear_f16_dev(outs[5].ref(), v2f(0, 0), srcs[1].ref(), v2f(0, 0), v2f(2.0, 2.0), None))
_lib.check(lib.cvs_stream_sync(None))
assert set_buffer(None) == 0
host = np.zeros((NWG, SLOTS), np.uint64)
_lib.check(lib.cvs_memcpy_d2h(host.ctypes.data, buf, host.nbytes, None))
t = host[host[:, 0] > 0].astype(np.int64)
t0 = t[:, 0].min()
names = ["start", "tap lists landed", "row range known", "line loop starts", "1st store issued", "8th store issued", "last store issued"]
if TILES:
    names = ["start", "ranges known", "taps+records landed", "rows in LDS", "barrier passed", "1st line stored", "(probe after start)", "last line stored"]
LAST = len(names) - 1
print("%s: %d workgroups; ticks of s_memtime (100 MHz constant clock on this part: 1 tick = 10 ns), kernel %d" % ("k_fir_tile_vh" if TILES else "k_fir_vh", len(t), lib.cvs_fir_last_kernel()))
print("launch span (first start -> last end): %d ticks" % (t[:, LAST].max() - t0))
print("%-20s %10s %10s %10s %10s" % ("point", "median", "p10", "p90", "max   (ticks after the launch's first start)"))
for k, n in enumerate(names):
    v = t[:, k] - t0
    print("%-20s %10d %10d %10d %10d" % (n, np.median(v), np.percentile(v, 10), np.percentile(v, 90), v.max()))
if TILES:
    r0 = t[:, 8].min()
    print("on the 100 MHz clock (1 tick = 10 ns), ticks after the launch's first workgroup started:")
    for k, n in ((0, "start"), (1, "ranges known"), (4, "barrier passed"), (5, "1st line stored"), (7, "last line stored")):
        v = t[:, 8 + k] - r0
        print("  %-18s p1 %6d  p10 %6d  median %6d  p90 %6d  p99 %6d  max %6d" % (n, np.percentile(v, 1), np.percentile(v, 10), np.median(v), np.percentile(v, 90), np.percentile(v, 99), v.max()))
    order = [0, 6, 1, 2, 3, 4, 5, 7]
    for base, what in ((0, "s_memtime"), (8, "s_memrealtime (100 MHz: 1 tick = 10 ns)")):
        print("per workgroup, medians of the differences, %s:" % what)
        for a, b in list(zip(order, order[1:])) + [(0, 7)]:
            d = t[:, base + b] - t[:, base + a]
            print("  %-20s -> %-20s %8d ticks (p10 %d, p90 %d)" % (names[a], names[b], np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
    sys.exit(0)
v = t[:, 7] - t[:, 7].min()
print("workgroup starts on the 100 MHz clock all XCDs share (1 tick = 10 ns), after the first: p10 %d  median %d  p90 %d  p99 %d  max %d" % (
    np.percentile(v, 10), np.median(v), np.percentile(v, 90), np.percentile(v, 99), v.max()))
print("per workgroup, medians of the differences:")
for a, b in [(k, k + 1) for k in range(LAST)] + [(0, LAST)]:
    d = t[:, b] - t[:, a]
    print("  %-18s -> %-18s %8d ticks (p10 %d, p90 %d)" % (names[a], names[b], np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
