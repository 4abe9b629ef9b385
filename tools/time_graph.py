#!/usr/bin/env python3
"""A launch-bound sequence with and without HIP graph replay: a ragged three-layer stack (node by node: ~14 short
kernels and a memset per frame) at several frame sizes, 200 iterations each, wall clock per iteration."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from canvas_amd import REC709_RGB_TO_YPBPR, _lib, synth  # noqa: E402
from canvas_amd.abi import HostFrame  # noqa: E402
from canvas_amd.device import DeviceFrame, chain_color_over  # noqa: E402

lib = _lib.load()
_lib.check(lib.cvs_init(0))
lib.init_half()
stream = lib.cvs_stream_create()
m = np.array(REC709_RGB_TO_YPBPR, np.float32)
for w, h in [(320, 180), (640, 360), (1920, 1080)]:
    full = (0, 0, w - 1, h - 1)
    wins = [full, (w // 16, h // 16, w // 2, h // 2), (w // 3, h // 20, w - 1, h // 2)]
    layers = []
    for k, win in enumerate(wins):
        f = synth.layer_frame(w, h, k, 0)
        layers.append(DeviceFrame.from_host(HostFrame(full, np.uint16, f.array, win)))
    out = DeviceFrame(full, np.uint16)
    run = lambda: chain_color_over([(out, layers)], m, _lib.LUT_REC709_TO_LINEAR_SCENE, _lib.LUT_NONE, stream)   # noqa: E731
    run()
    lib.cvs_stream_sync(stream)
    assert lib.cvs_chain_last_was_fused() == 0
    _lib.check(lib.cvs_graph_begin(stream))
    run()
    graph = lib.cvs_graph_end(stream)
    assert graph, _lib.last_error()
    res = {}
    for name, fn in [("direct", run), ("graph", lambda: _lib.check(lib.cvs_graph_launch(graph, stream)))]:
        for _ in range(20):
            fn()
        lib.cvs_stream_sync(stream)
        t0 = time.perf_counter()
        for _ in range(200):
            fn()
        lib.cvs_stream_sync(stream)
        res[name] = (time.perf_counter() - t0) / 200 * 1e3
    lib.cvs_graph_destroy(graph)
    print("%dx%d ragged 3-layer stack, node by node: direct %.4f ms, graph replay %.4f ms per frame (%.2fx)" % (w, h, res["direct"], res["graph"], res["direct"] / res["graph"]))
