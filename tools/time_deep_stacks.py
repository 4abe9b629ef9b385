#!/usr/bin/env python3
"""Workspace stacks of 2..8 layers at 3840x2160 (plain over, f16 in and out) through the chain kernel (1-4 layers:
chain_ops.hip, 5-8: chain_deep_ops.hip).  8 frames per call; GB/s against 8 B per layer pixel + 8 B per output pixel."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from canvas_amd import _lib, synth  # noqa: E402
from canvas_amd.device import DeviceFrame, chain_color_over  # noqa: E402

lib = _lib.load()
_lib.check(lib.cvs_init(0))
lib.init_half()
stream = lib.cvs_stream_create()
w, h, n = 3840, 2160, 8
full = (0, 0, w - 1, h - 1)
layers = [DeviceFrame.from_host(synth.layer_frame(w, h, k, 0)) for k in range(8)]
outs = [DeviceFrame(full, np.uint16) for _ in range(n)]
e0, e1 = lib.cvs_event_create(), lib.cvs_event_create()
for nl in range(2, 9):
    jobs = [(outs[i], [layers[(i + k) % 8] if k else layers[0] for k in range(nl)]) for i in range(n)]
    chain_color_over(jobs, None, _lib.LUT_NONE, _lib.LUT_NONE, stream)
    lib.cvs_stream_sync(stream)
    fused = lib.cvs_chain_last_was_fused()
    ts = []
    for _ in range(7):
        lib.cvs_event_record(e0, stream)
        chain_color_over(jobs, None, _lib.LUT_NONE, _lib.LUT_NONE, stream)
        lib.cvs_event_record(e1, stream)
        lib.cvs_stream_sync(stream)
        ts.append(lib.cvs_event_elapsed_ms(e0, e1))
    ms = sorted(ts)[3] / n
    print("%d layers: %.4f ms per frame, %d B/px -> %.0f GB/s (one launch: %s)" % (nl, ms, 8 * (nl + 1), w * h * 8 * (nl + 1) / ms / 1e6, bool(fused)))
