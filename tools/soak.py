#!/usr/bin/env python3
"""Soak: the three hot kernels re-run thousands of times on the same inputs; every 40th result is downloaded and must
equal the first one bit for bit (a rare ordering bug in a software pipeline shows up as a handful of wrong pixels once
in many launches, not in a parity test that runs each case once)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from canvas_amd import REC709_RGB_TO_YPBPR, _lib, synth  # noqa: E402
from canvas_amd.device import DeviceFrame, chain_color_over  # noqa: E402
from canvas_amd.stream import GraphStream  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
lib = _lib.load()
_lib.check(lib.cvs_init(0))
lib.init_half()
stream = lib.cvs_stream_create()
w, h = 3840, 2160
full = (0, 0, w - 1, h - 1)
m = np.array(REC709_RGB_TO_YPBPR, np.float32)
sets = []
for g in range(4):
    layers = [DeviceFrame.from_host(synth.layer_frame(w, h, k, g)) for k in range(2)]
    sets.append((DeviceFrame(full, np.uint16), layers))
graph = GraphStream(w, h, ring=2)
taps = synth.gaussian_taps(9, 1.5)
small = DeviceFrame((0, 0, w // 2 - 1, h // 2 - 1), np.uint16)
f32p = taps.ctypes.data_as(C.POINTER(C.c_float))
# the sweep resampler (4K -> 1536x864 Lanczos3 behind a 1-tap blur) and a 21-tap blur
one = np.array([1.0], np.float32)
onep = one.ctypes.data_as(C.POINTER(C.c_float))
odd = DeviceFrame((0, 0, int(w * 0.4) - 1, int(h * 0.4) - 1), np.uint16)
taps21 = synth.gaussian_taps(21, 3.5)
t21p = taps21.ctypes.data_as(C.POINTER(C.c_float))
wide = DeviceFrame(full, np.uint16)
# the channel-pair sweep with 16 slots (1.5x enlargement) and its 12-tap form (0.75x); a deep stack on the chain kernel
big = DeviceFrame((0, 0, int(w * 1.5) - 1, int(h * 1.5) - 1), np.uint16)
mid = DeviceFrame((0, 0, int(w * 0.75) - 1, int(h * 0.75) - 1), np.uint16)
deep = DeviceFrame(full, np.uint16)
# the reference's scaler in one launch: vertical pass first (2x, 0.5x) and horizontal first (0.75 x 1.5)
from canvas_amd.abi import v2f  # noqa: E402
up2 = DeviceFrame((0, 0, 2 * w - 1, 2 * h - 1), np.uint16)
half = DeviceFrame((0, 0, w // 2 - 1, h // 2 - 1), np.uint16)
anam = DeviceFrame((0, 0, int(w * 0.75) - 1, int(h * 1.5) - 1), np.uint16)
# the tile form of the vertical-first scaler (round 4): 1080p -> 4K halfs, single and three frames per launch (the batch entry,
# halfs on the strips), and floats
hd = [DeviceFrame.from_host(synth.layer_frame(w // 2, h // 2, 1, g)) for g in range(3)]
hd32 = DeviceFrame((0, 0, w // 2 - 1, h // 2 - 1), np.float32)
_lib.check(lib.cvs_frame_f16_to_f32_dev(hd32.ref(), hd[0].ref(), None))
_lib.check(lib.cvs_stream_sync(None))
up4k = DeviceFrame(full, np.uint16)
up4k32 = DeviceFrame(full, np.float32)
upb = [DeviceFrame(full, np.uint16) for _ in range(3)]
_tab16 = lambda fr: (C.POINTER(_lib.rgba_frame_f16_t) * len(fr))(*[C.pointer(f.c) for f in fr])
upb_t, hd_t = _tab16(upb), _tab16(hd)


def once():
    chain_color_over([sets[i % 4] for i in range(8)], m, _lib.LUT_REC709_TO_LINEAR_SCENE, _lib.LUT_NONE, stream)
    graph.render(0, stream)
    graph.render(1, stream)
    _lib.check(lib.cvs_blur_lanczos_f16_dev(small.ref(), sets[0][1][1].ref(), f32p, 9, C.c_float(0.5), C.c_float(0.5), 3, stream))
    _lib.check(lib.cvs_blur_lanczos_f16_dev(odd.ref(), sets[1][1][1].ref(), onep, 1, C.c_float(0.4), C.c_float(0.4), 3, stream))
    _lib.check(lib.cvs_fir_blur_f16_dev(wide.ref(), sets[2][1][1].ref(), t21p, 21, stream))
    _lib.check(lib.cvs_resample_lanczos_f16_dev(big.ref(), sets[3][1][1].ref(), C.c_float(1.5), C.c_float(1.5), 3, stream))
    _lib.check(lib.cvs_resample_lanczos_f16_dev(mid.ref(), sets[0][1][0].ref(), C.c_float(0.75), C.c_float(0.75), 3, stream))
    chain_color_over([(deep, [sets[i % 4][1][i % 2] for i in range(6)])], None, _lib.LUT_NONE, _lib.LUT_NONE, stream)
    _lib.check(lib.cvs_scale_bilinear_f16_dev(up2.ref(), v2f(0, 0), sets[1][1][0].ref(), v2f(0, 0), v2f(2.0, 2.0), stream))
    _lib.check(lib.cvs_scale_bilinear_f16_dev(half.ref(), v2f(0, 0), sets[2][1][0].ref(), v2f(0, 0), v2f(0.5, 0.5), stream))
    _lib.check(lib.cvs_scale_bilinear_f16_dev(anam.ref(), v2f(0, 0), sets[3][1][0].ref(), v2f(0, 0), v2f(0.75, 1.5), stream))
    _lib.check(lib.cvs_scale_bilinear_f16_dev(up4k.ref(), v2f(0, 0), hd[0].ref(), v2f(0, 0), v2f(2.0, 2.0), stream))
    _lib.check(lib.cvs_scale_bilinear_f32_dev(up4k32.ref(), v2f(0, 0), hd32.ref(), v2f(0, 0), v2f(2.0, 2.0), stream))
    _lib.check(lib.cvs_scale_bilinear_f16_batch_dev(upb_t, v2f(0, 0), hd_t, v2f(0, 0), v2f(2.0, 2.0), 3, stream))


def snapshot():
    _lib.check(lib.cvs_stream_sync(stream))
    return [s[0].download().array.copy() for s in sets] + [graph.slots[0]["out"].download().array.copy(), graph.slots[1]["out"].download().array.copy(),
                                                        small.download().array.copy(), odd.download().array.copy(), wide.download().array.copy(),
                                                        big.download().array.copy(), mid.download().array.copy(), deep.download().array.copy(),
                                                        up2.download().array.copy(), half.download().array.copy(), anam.download().array.copy(),
                                                        up4k.download().array.copy(), up4k32.download().array.copy()] + [u.download().array.copy() for u in upb]


once()
first = snapshot()
t0, n, checks = time.perf_counter(), 0, 0
while time.perf_counter() - t0 < seconds:
    for _ in range(40):
        once()
    n += 40
    now = snapshot()
    for a, b in zip(first, now):
        if not np.array_equal(a, b):
            bad = int((a != b).sum())
            print("MISMATCH after %d iterations: %d values differ" % (n, bad))
            sys.exit(1)
    checks += 1
    if checks % 50 == 0:                      # (a silent GPU command is taken for hung after a few minutes)
        print("  %d iterations, %d compares, %.0f s" % (n, checks, time.perf_counter() - t0), flush=True)
print("soak ok: %d iterations (%d launches of the chain over 8 frames, %d config-5 frames, %d config-3 frames, as many 0.4x / 0.75x / 1.5x resamples, 21-tap blurs, 6-layer stacks and scaler calls at 2x, 0.5x and 0.75 x 1.5, 1080p -> 4K on the tile kernel in halfs and floats and three frames in one launch), %d full compares, %.1f s"
      % (n, n, 2 * n, n, checks, time.perf_counter() - t0))
