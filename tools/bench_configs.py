#!/usr/bin/env python3
"""Throughput of the other BASELINE configs' kernels on device-resident frames (1 GPU).
Not the driver's bench line (that is bench.py = config 2); evidence for DESIGN.md section 4.

  config 3: 3840x2160 -> 9-tap separable Gaussian (sigma 1.5) -> Lanczos3 to 1920x1080
  config 4: 7680x4320 3-layer alpha-over stack (f16 layers, f32 over, f16 out)
  unfused : config 2 node by node (colour x2, widen x2, over, narrow), for comparison with the fused kernel
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from canvas_amd import REC709_RGB_TO_YPBPR, _lib, synth  # noqa: E402
from canvas_amd.device import DeviceFrame, chain_color_over  # noqa: E402


def timed(lib, stream, fn, reps):
    e0, e1 = lib.cvs_event_create(), lib.cvs_event_create()
    fn()
    _lib.check(lib.cvs_stream_sync(stream))
    ts = []
    for _ in range(reps):
        lib.cvs_event_record(e0, stream)
        fn()
        lib.cvs_event_record(e1, stream)
        _lib.check(lib.cvs_stream_sync(stream))
        ts.append(lib.cvs_event_elapsed_ms(e0, e1))
    return float(np.median(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--which", default="3,4,unfused")
    args = ap.parse_args()
    lib = _lib.load()
    _lib.check(lib.cvs_init(0))
    lib.init_half()
    stream = lib.cvs_stream_create()
    out = {}
    f32p = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731

    if "3" in args.which.split(","):
        w, h = 3840, 2160
        src16 = DeviceFrame.from_host(synth.layer_frame(w, h, 0, 0))
        src = DeviceFrame((0, 0, w - 1, h - 1), np.float32)
        _lib.check(lib.cvs_frame_f16_to_f32_dev(src.ref(), src16.ref(), stream))
        blurred = DeviceFrame((0, 0, w - 1, h - 1), np.float32)
        small = DeviceFrame((0, 0, w // 2 - 1, h // 2 - 1), np.float32)
        taps = synth.gaussian_taps(9, 1.5)
        t_blur = timed(lib, stream, lambda: _lib.check(lib.cvs_fir_blur_f32_dev(blurred.ref(), src.ref(), f32p(taps), 9, stream)), args.reps)
        t_scale = timed(lib, stream, lambda: _lib.check(lib.cvs_resample_lanczos_f32_dev(small.ref(), blurred.ref(), C.c_float(0.5), C.c_float(0.5), 3, stream)), args.reps)
        px = w * h
        # algorithmic bytes in the f32 form these kernels work in: blur 16 r + 16 w per px; scale 16 r + 16/4 w
        out["config3"] = {"blur_ms": t_blur, "scale_ms": t_scale, "Mpx_per_s_input": px / ((t_blur + t_scale) * 1e-3) / 1e6,
                          "blur_GBps": px * 32 / (t_blur * 1e-3) / 1e9, "scale_GBps": px * 20 / (t_scale * 1e-3) / 1e9,
                          "note": "blur_ms / scale_ms: f32 frames in and out, one launch each (register-window FIR kernel, taps as kernel arguments); pipeline_f16_ms: cvs_blur_lanczos_f16_dev, f16 in, f32 frame between the two launches, f16 out"}
        # the config as BASELINE states it: f16 in, f16 out, f32 in between (two fused launches)
        out16 = DeviceFrame((0, 0, w // 2 - 1, h // 2 - 1), np.uint16)
        t_pipe = timed(lib, stream, lambda: _lib.check(lib.cvs_blur_lanczos_f16_dev(out16.ref(), src16.ref(), f32p(taps), 9, C.c_float(0.5), C.c_float(0.5), 3, stream)), args.reps)
        per_node = px * 26          # 8 r + 8 w (blur) + 8 r + 8/4 w (scale) per input px
        fused_lb = px * 8 + px * 2  # read the f16 source once, write the f16 quarter-size result once
        actual = px * 8 + px * 16 + px * 16 + px * 2   # this implementation: f16 r, f32 w, f32 r, f16 w
        out["config3"].update({"pipeline_f16_ms": t_pipe, "pipeline_Mpx_per_s_input": px / (t_pipe * 1e-3) / 1e6,
                               "pipeline_GBps_per_node_denominator": per_node / (t_pipe * 1e-3) / 1e9,
                               "pipeline_GBps_fused_lower_bound_denominator": fused_lb / (t_pipe * 1e-3) / 1e9,
                               "pipeline_GBps_actual_traffic_model": actual / (t_pipe * 1e-3) / 1e9})
        out16.free()
        for d in (src16, src, blurred, small):
            d.free()

    if "4" in args.which.split(","):
        w, h, nl = 7680, 4320, 3
        ident = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1], np.float32)
        ring = []
        for g in range(2):
            layers = []
            for k in range(nl):
                d = DeviceFrame((0, 0, w - 1, h - 1), np.uint16)
                d.upload(synth.layer_pixels(w, h, k, g))
                layers.append(d)
            ring.append((DeviceFrame((0, 0, w - 1, h - 1), np.uint16), layers))
        t = timed(lib, stream, lambda: chain_color_over(ring, ident, _lib.LUT_NONE, _lib.LUT_NONE, stream), args.reps)
        px = 2 * w * h
        out["config4"] = {"ms_per_2_frames": t, "Mpx_per_s": px / (t * 1e-3) / 1e6, "GBps": px * 8 * (nl + 1) / (t * 1e-3) / 1e9,
                          "frac_of_8TBps": px * 8 * (nl + 1) / (t * 1e-3) / 8e12,
                          "note": "fused chain kernel, 3 layers, identity matrix, no LUT; 32 B/px algorithmic"}
        for o, ls in ring:
            o.free()
            for l in ls:
                l.free()

    if "unfused" in args.which.split(","):
        w, h = 3840, 2160
        m = np.array(REC709_RGB_TO_YPBPR, np.float32)
        full = (0, 0, w - 1, h - 1)
        layers = [DeviceFrame.from_host(synth.layer_frame(w, h, k, 0)) for k in range(2)]
        graded = [DeviceFrame(full, np.uint16) for _ in range(2)]
        f32 = [DeviceFrame(full, np.float32) for _ in range(2)]
        res = DeviceFrame(full, np.uint16)

        def node_by_node():
            for k in range(2):
                _lib.check(lib.cvs_copy_frame_f16_dev(graded[k].ref(), layers[k].ref(), stream))
                _lib.check(lib.cvs_color_matrix_f16_dev(graded[k].ref(), f32p(m), 0, -1, stream))
                _lib.check(lib.cvs_frame_f16_to_f32_dev(f32[k].ref(), graded[k].ref(), stream))
            _lib.check(lib.cvs_mix_over_f32_dev(f32[0].ref(), f32[1].ref(), C.c_float(1.0), stream))
            _lib.check(lib.cvs_frame_f32_to_f16_dev(res.ref(), f32[0].ref(), stream))

        t = timed(lib, stream, node_by_node, args.reps)
        out["config2_node_by_node"] = {"ms_per_frame": t, "Mpx_per_s": w * h / (t * 1e-3) / 1e6,
                                       "note": "8 kernels per frame, f32 intermediates in HBM: 2x(copy 16 + colour 16 + widen 24) + over 48 + narrow 24 = 184 B/px"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
